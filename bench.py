#!/usr/bin/env python3
"""MSDA forward+backward micro-benchmark (BASELINE.json metric: Gsamples/s and % of HBM peak).

A *step* is one forward + one backward pass of the hot path over one batch of synthetic
encoder inputs, through the drop-in boundary (MultiScaleDeformableAttention -> C ABI -> HIP).
Default workload = the configuration BASELINE.md quotes the target on ("C3"): the Swin-B
pixel-decoder shape, bs=32, 1024^2 input -> levels 32^2/64^2/128^2 (low-res first), 8 heads x
32 channels, 4 points, fp32 (the reference's arithmetic), model-like sampling locations
(SURVEY.md section 8(d)); inputs are resident in HBM before the timed region.  At one GPU the
line's `detail.other_workloads` also carries short runs of C3 in bf16 (BASELINE.json
configs[2]) and of C2 (configs[1], bs=16 fp32).

Multi-GPU: the op shards by image only, so every rank runs the same per-GPU batch on its own
data with no data-path collective (weak scaling).  Ranks are started by torch.distributed.run;
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts them itself
as fresh child processes (before this process has touched the GPU) and exits with their code.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak

WORKLOADS = {
    # name: (batch per GPU, image size)
    "C2": (16, 1024),          # Swin-T pixel-decoder shape (BASELINE.json configs[1])
    "C3": (32, 1024),          # Swin-B shape (configs[2]; the shape the 50 % target is stated on)
    "C5": (2, 1536),           # Swin-L 1536^2, per-GPU batch of the 8-GPU run
    "C5x16": (16, 1536),       # same, single-GPU stress
    "tiny": (2, 256),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--dist", default="model", choices=["model", "uniform"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: value/output/grad_output in bf16, everything else fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the short C3-bf16 / C2 runs reported under detail.other_workloads")
    ap.add_argument("--cpu-images", type=int, default=2)
    ap.add_argument("--cpu-reps", type=int, default=5)
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this
    process, which has not imported torch or touched the GPU, and hand back their exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def host_threads():
    """Threads for the CPU baseline: the cores this process may run on (its affinity mask, i.e.
    the box's CPU share), unless OMP_NUM_THREADS already says so."""
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if env > 0:
        return env
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(torch, shapes, dist, images, reps):
    """Times the PyTorch restatement of the reference's CPU path (oracle/msda_torch.py) --
    the checker, run here only as the reported CPU baseline -- fwd + autograd bwd, fp32."""
    from instaorder_amd import synthetic
    from oracle import msda_torch
    cores = host_threads()
    torch.set_num_threads(cores)
    inp = synthetic.make_inputs(images, shapes, dist=dist, seed=0)
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        msda_torch.msda_forward_backward_torch(inp["value"], inp["spatial_shapes"],
                                               inp["sampling_locations"], inp["attention_weights"],
                                               inp["grad_output"])
        best = min(best, time.perf_counter() - t0)
    samples = synthetic.num_samples(images, shapes)
    return {"value": samples / best / 1e9, "unit": "Gsamples/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d image(s) of the same workload, fwd+autograd bwd fp32, best of %d "
                      "(oracle/msda_torch.py: grid_sample restatement of "
                      "ms_deform_attn_core_pytorch), %.3f s per pass; %d torch threads "
                      "(OMP_NUM_THREADS=%s, affinity %d, os.cpu_count() %d)"
                      % (images, reps, best, torch.get_num_threads(),
                         os.environ.get("OMP_NUM_THREADS", "unset"),
                         len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else -1,
                         os.cpu_count() or -1)}


def bytes_per_sample(synthetic, dtype):
    if dtype == "bf16":      # value, output, grad_output AND grad_value in bf16: 22.67 + 40.00 = 62.67 B/sample
        return synthetic.algorithmic_bytes_per_sample(value_bytes=2)
    return synthetic.algorithmic_bytes_per_sample()


class Workload:
    """Inputs of one workload resident on the GPU and the step over them."""

    def __init__(self, torch, MSDA, synthetic, name, dist, dtype, seed):
        self.torch, self.MSDA = torch, MSDA
        self.name, self.dist, self.dtype = name, dist, dtype
        self.batch, self.image = WORKLOADS[name]
        self.shapes = synthetic.encoder_shapes(self.image)
        inp = synthetic.make_inputs(self.batch, self.shapes, dist=dist, seed=seed, device="cuda")
        self.v, self.ss, self.lsi = inp["value"], inp["spatial_shapes"], inp["level_start_index"]
        self.loc, self.aw, self.go = inp["sampling_locations"], inp["attention_weights"], inp["grad_output"]
        if dtype == "bf16":
            self.v, self.go = self.v.to(torch.bfloat16), self.go.to(torch.bfloat16)
        self.samples = synthetic.num_samples(self.batch, self.shapes)
        self.fwd_b, self.bwd_b = bytes_per_sample(synthetic, dtype)

    def forward(self):
        return self.MSDA.ms_deform_attn_forward(self.v, self.ss, self.lsi, self.loc, self.aw, 128)

    def backward(self):
        return self.MSDA.ms_deform_attn_backward(self.v, self.ss, self.lsi, self.loc, self.aw, self.go, 128)

    def step(self):
        return self.forward(), self.backward()

    def event_times(self, steps):
        """Average forward / backward call durations (ms) from HIP events on torch's current
        stream -- the stream the C ABI launches on."""
        torch = self.torch
        fwd_ms, bwd_ms = [], []
        for _ in range(steps):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            self.forward()
            e1.record()
            self.backward()
            e2.record()
            e2.synchronize()
            fwd_ms.append(e0.elapsed_time(e1))
            bwd_ms.append(e1.elapsed_time(e2))
        return sum(fwd_ms) / len(fwd_ms), sum(bwd_ms) / len(bwd_ms)

    def describe(self):
        return ("%s: MSDeformAttn fwd+bwd, bs=%d/GPU, %d^2 input, levels %s, M=8, D=32, P=4, "
                "Lq=S=%d, %s locations, %s"
                % (self.name, self.batch, self.image, "/".join("%dx%d" % s for s in self.shapes),
                   sum(h * w for h, w in self.shapes), self.dist,
                   "fp32" if self.dtype == "f32" else "bf16 value/output/grad_output/grad_value (62.67 B/sample), fp32 accumulate"))

    def fractions(self, fwd_ms, bwd_ms):
        s = self.samples
        fwd_gbs = s * self.fwd_b / (fwd_ms * 1e-3) / 1e9
        bwd_gbs = s * self.bwd_b / (bwd_ms * 1e-3) / 1e9
        both_gbs = s * (self.fwd_b + self.bwd_b) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9
        return fwd_gbs, bwd_gbs, both_gbs


class FusedWorkload(Workload):
    """The same workload through the opt-in fused boundary: the op plus the softmax
    and the sampling-location arithmetic of its caller, from the raw offsets and logits."""

    def __init__(self, torch, MSDA, synthetic, name, dist, dtype, seed):
        super().__init__(torch, MSDA, synthetic, name, dist, dtype, seed)
        from instaorder_amd import ext
        self.ext = ext
        n, lq, m, nl, p, _ = self.loc.shape
        self.ref = synthetic.reference_points(self.shapes).to(self.loc.device)[None, :, None, :].expand(n, lq, nl, 2).contiguous()
        wh = torch.tensor([(w, h) for h, w in self.shapes], dtype=torch.float32, device=self.loc.device)
        self.off = ((self.loc - self.ref[:, :, None, :, None, :]) * wh[None, None, None, :, None, :]).contiguous()
        self.logits = torch.log(self.aw.clamp_min(1e-20)).reshape(n, lq, m, nl * p).contiguous()
        del self.loc, self.aw

    def forward(self):
        return self.ext.ms_deform_attn_fused_forward(self.v, self.ss, self.lsi, self.ref, self.off, self.logits)

    def backward(self):
        return self.ext.ms_deform_attn_fused_backward(self.v, self.ss, self.lsi, self.ref, self.off, self.logits, self.go)

    def describe(self):
        return super().describe() + "; FUSED boundary (row f3, opt-in): softmax and location arithmetic of the caller inside the calls"


def short_run(torch, MSDA, synthetic, name, dtype, steps=10, warmup=3, fused=False):
    w = (FusedWorkload if fused else Workload)(torch, MSDA, synthetic, name, "model", dtype, seed=0)
    for _ in range(warmup):
        w.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    fwd_ms, bwd_ms = w.event_times(steps)
    _, _, both = w.fractions(fwd_ms, bwd_ms)
    return {"workload": w.describe(), "gsamples_s": round(w.samples / wall / 1e9, 3),
            "ms_per_step": round(wall * 1e3, 4), "fwd_ms": round(fwd_ms, 4), "bwd_ms": round(bwd_ms, 4),
            "fwd_bwd_hbm_frac": round(both / HBM_PEAK_GBS, 5), "steps": steps}


def rehearsal(args, world, rank):
    """MSDA_BENCH_REHEARSAL=1 on a GPU-less host: the launch, rendezvous (gloo), barrier,
    max-over-ranks and JSON plumbing with NO compute -- a test of the control path only."""
    from instaorder_amd import distributed as D
    D.init("gloo")
    D.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps)
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "msda_fwd_bwd_throughput", "value": None, "unit": "Gsamples/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "rehearsal": True,
                          "data": "none (control-path rehearsal without a GPU; no kernels ran)"}), flush=True)
    D.shutdown()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    if not torch.cuda.is_available():
        if os.environ.get("MSDA_BENCH_REHEARSAL") == "1":
            return rehearsal(args, world, rank)
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    # one GPU per rank; MSDA_BENCH_BACKEND=gloo lets several ranks rehearse the launch on ONE GPU
    # (RCCL refuses two ranks on a device) -- a rehearsal of the control path, not a measurement
    backend = os.environ.get("MSDA_BENCH_BACKEND", "nccl")
    if backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    import MultiScaleDeformableAttention as MSDA
    from instaorder_amd import distributed as D
    from instaorder_amd import ext, synthetic
    D.init(backend)           # no-op for one rank; RCCL process group otherwise

    w = Workload(torch, MSDA, synthetic, args.workload, args.dist, args.dtype, seed=rank)
    barrier = D.barrier       # rendezvous of the ranks + torch.cuda.synchronize()

    for _ in range(args.warmup):
        w.step()
    launches0 = ext.launch_count()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w.step()
    barrier()
    elapsed = time.perf_counter() - t0
    assert ext.launch_count() >= launches0 + 2 * args.steps, "HIP path did not run"
    elapsed = D.max_over_ranks(elapsed)

    # Per-call durations with HIP events, separate from the wall-clock region above.
    fwd_avg, bwd_avg = w.event_times(args.steps)

    if rank != 0:
        D.shutdown()
        return

    samples = w.samples
    ms_per_step = elapsed / args.steps * 1e3
    value = samples * world / (elapsed / args.steps) / 1e9
    # dominant kernel = the backward call: the binning passes and the tile kernel on the same
    # stream, all inside the event bracket
    fwd_gbs, bwd_gbs, both_gbs = w.fractions(fwd_avg, bwd_avg)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("%s/%s/%s" % (args.workload, args.dist, args.dtype), {}).get(
                "backward_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    result = {
        "metric": "msda_fwd_bwd_throughput",
        "value": round(value, 4),
        "unit": "Gsamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": w.describe(), "samples_per_step_per_gpu": samples,
                   "parallelism": "batch-sharded x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "msda_backward call (binning passes + tile kernel)",
                     "achieved": round(bwd_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(bwd_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(samples * w.bwd_b),
                     "avg_launch_ms": round(bwd_avg, 4)},
        "detail": {"fwd_ms": round(fwd_avg, 4), "bwd_ms": round(bwd_avg, 4),
                   "fwd_gsamples_s": round(samples / fwd_avg / 1e6, 3),
                   "bwd_gsamples_s": round(samples / bwd_avg / 1e6, 3),
                   "fwd_hbm_frac": round(fwd_gbs / HBM_PEAK_GBS, 5),
                   "fwd_bwd_hbm_frac": round(both_gbs / HBM_PEAK_GBS, 5),
                   "fwd_bwd_gsamples_s_events": round(samples / (fwd_avg + bwd_avg) / 1e6, 3)},
    }
    if world == 1 and not args.no_extras:
        del w
        torch.cuda.empty_cache()
        extras = []
        for name, dtype in (("C3", "bf16"), ("C2", "f32")):
            if (name, dtype) != (args.workload, args.dtype):
                extras.append(short_run(torch, MSDA, synthetic, name, dtype))
                torch.cuda.empty_cache()
        try:      # the opt-in fused boundary at the headline shape; never at the expense of the line itself
            extras.append(short_run(torch, MSDA, synthetic, args.workload, args.dtype, fused=True))
        except Exception as e:  # noqa: BLE001
            extras.append({"workload": "fused boundary", "error": str(e)[:200]})
        torch.cuda.empty_cache()
        result["detail"]["other_workloads"] = extras
    if not args.no_cpu_baseline and world == 1:
        shapes = synthetic.encoder_shapes(WORKLOADS[args.workload][1])
        result["cpu_baseline"] = cpu_baseline(torch, shapes, args.dist, args.cpu_images, args.cpu_reps)
    print(json.dumps(result), flush=True)
    D.shutdown()


if __name__ == "__main__":
    main()
