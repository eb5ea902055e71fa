#!/usr/bin/env python3
"""Training-step harness for the MSDA encoder: synthetic
multi-scale features -> 6-layer MSDeformAttn encoder (adapters on) -> scalar loss -> backward
-> DistributedDataParallel gradient all-reduce (RCCL over xGMI on GPUs, gloo on CPU).

The global batch is sharded by image over the ranks, as the reference's detectron2 launch does
(train_net.py:404-414, SOLVER.IMS_PER_BATCH split evenly).  With --freeze-base only the
adapters train, like InstaFormer^{o,d} (mask2former/maskformer_model.py:167-182): the first
layer's op then needs no backward, giving the reference's 6 forward + 5 backward MSDA calls.
detectron2 is not available offline, so the backbone / M2F decoder / criterion of the full model
are NOT part of this harness: it times the encoder stack, the part this repository owns.

  python train_bench.py --image 1024 --global-batch 16 --steps 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
         train_bench.py --gpus 2 ...

Prints one JSON line on rank 0 (imgs/s over the whole job)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from instaorder_amd import distributed as D                      # noqa: E402
from instaorder_amd import synthetic                              # noqa: E402
from instaorder_amd.encoder import MSDeformAttnTransformerEncoderOnly, PositionEmbeddingSine  # noqa: E402


def build_encoder(levels=3, freeze_base=True, seed=0):
    torch.manual_seed(seed)
    enc = MSDeformAttnTransformerEncoderOnly(d_model=256, nhead=8, num_encoder_layers=6,
                                             dim_feedforward=1024, dropout=0.0,
                                             num_feature_levels=levels, enc_n_points=4,
                                             use_adapters=True)
    if freeze_base:
        for name, p in enc.named_parameters():
            p.requires_grad_("adapter" in name)
    return enc


def launch_ranks(args):
    """`python train_bench.py --gpus N` without a launcher: start the N ranks as children of this
    process (which has imported torch but made no GPU call) and hand back their exit code --
    the single-node form of the reference's detectron2 `launch(main, num_gpus, ...)`
    (train_net.py:404-414)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def run(args):
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE", "unset")))
    world, rank, local = D.init("gloo" if args.cpu else None)
    device = torch.device("cuda", local) if torch.cuda.is_available() and not args.cpu else torch.device("cpu")
    shapes = synthetic.encoder_shapes(args.image)
    enc = build_encoder(len(shapes), args.freeze_base).to(device)
    if args.fused:      # row f3: softmax + sampling-location arithmetic inside the MSDA kernels
        for mod in enc.modules():
            if hasattr(mod, "fused_prologue"):
                mod.fused_prologue = True
    model = enc
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        model = DDP(enc, device_ids=[local] if device.type == "cuda" else None)
    opt = torch.optim.AdamW([p for p in enc.parameters() if p.requires_grad], lr=1e-4)
    lo, hi = D.shard_batch(args.global_batch, world, rank)
    g = torch.Generator().manual_seed(1234)
    feats = [torch.randn((args.global_batch, 256, h, w), generator=g)[lo:hi].to(device) for h, w in shapes]
    pe = PositionEmbeddingSine(128, normalize=True)
    pos = [pe(x) for x in feats]

    def step():
        opt.zero_grad(set_to_none=True)
        memory, _, _ = model(feats, pos)
        loss = memory.float().pow(2).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({
            "metric": "msda_encoder_train_step", "value": round(args.global_batch * args.steps / elapsed, 3),
            "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "scaling": "strong",
            "config": {"workload": "6-layer MSDeformAttn encoder (d=256, 8 heads, 4 points, adapters), "
                                   "%d^2 input, levels %s, global batch %d, %s"
                                   % (args.image, "/".join("%dx%d" % s for s in shapes), args.global_batch,
                                      "adapters only trainable" if args.freeze_base else "all trainable"),
                       "device": device.type, "final_loss": float(loss.detach())}}), flush=True)
    D.shutdown()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--image", type=int, default=1024)
    ap.add_argument("--global-batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--all-trainable", dest="freeze_base", action="store_false")
    ap.add_argument("--fused", action="store_true",
                    help="MSDeformAttn.fused_prologue: the fused entry points (SURVEY section 8, row f3)")
    run(ap.parse_args())
