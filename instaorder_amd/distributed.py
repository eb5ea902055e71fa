"""One process per GPU, batch-sharded replicas (the op has no cross-image dependence, so the
data path needs no collective).  torch.distributed backend "nccl" is
RCCL on ROCm; "gloo" is used by the CPU tests.  Ranks come from torch.distributed.run's env."""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None):
    """Initialises the default process group when WORLD_SIZE > 1.  Returns (world, rank, local)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return world, rank, local


def barrier():
    """Rendezvous of all ranks, then drain this rank's GPU queue (both sides of a timed region)."""
    if dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(seconds):
    """The slowest rank's time: a job is as fast as its last replica."""
    if not dist.is_initialized():
        return float(seconds)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_batch(global_batch, world, rank):
    """Images of a global batch owned by `rank` (contiguous, remainder to the low ranks)."""
    base, extra = divmod(global_batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
