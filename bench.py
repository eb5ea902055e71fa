#!/usr/bin/env python3
"""MSDA forward+backward micro-benchmark (BASELINE.json metric: Gsamples/s and % of HBM peak).

A *step* is one forward + one backward pass of the hot path over one batch of synthetic
encoder inputs, through the drop-in boundary (MultiScaleDeformableAttention -> C ABI -> HIP).
Default workload = BASELINE.json configs[1] ("C2"): bs=16, 1024^2 input -> levels
32^2/64^2/128^2 (low-res first), 8 heads x 32 channels, 4 points, fp32, model-like sampling
locations; inputs are resident in HBM before the timed region.

Multi-GPU: the op shards by image only, so every rank runs the same per-GPU batch on its own
data with no data-path collective (weak scaling); ranks are launched by torch.distributed.run.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak

WORKLOADS = {
    # name: (batch per GPU, image size)
    "C2": (16, 1024),          # Swin-T pixel-decoder shape (BASELINE.json configs[1])
    "C3": (32, 1024),          # Swin-B shape, fp32
    "C5": (2, 1536),           # Swin-L 1536^2, per-GPU batch of the 8-GPU run
    "C5x16": (16, 1536),       # same, single-GPU stress
    "tiny": (2, 256),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--dist", default="model", choices=["model", "uniform"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: value/output/grad_output in bf16, everything else fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=2)
    ap.add_argument("--cpu-reps", type=int, default=5)
    return ap.parse_args()


def cpu_baseline(shapes, dist, images, reps):
    """Times the PyTorch restatement of the reference's CPU path (oracle/msda_torch.py) --
    the checker, run here only as the reported CPU baseline -- fwd + autograd bwd, fp32."""
    from instaorder_amd import synthetic
    from oracle import msda_torch
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
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
                      "ms_deform_attn_core_pytorch), %.3f s per pass" % (images, reps, best)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 "
                             "--nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
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
    dist_on = world > 1

    batch, image = WORKLOADS[args.workload]
    shapes = synthetic.encoder_shapes(image)
    inp = synthetic.make_inputs(batch, shapes, dist=args.dist, seed=rank, device="cuda")
    v, ss, lsi = inp["value"], inp["spatial_shapes"], inp["level_start_index"]
    loc, aw, go = inp["sampling_locations"], inp["attention_weights"], inp["grad_output"]
    if args.dtype == "bf16":
        v, go = v.to(torch.bfloat16), go.to(torch.bfloat16)
    samples = synthetic.num_samples(batch, shapes)

    def step():
        out = MSDA.ms_deform_attn_forward(v, ss, lsi, loc, aw, 128)
        grads = MSDA.ms_deform_attn_backward(v, ss, lsi, loc, aw, go, 128)
        return out, grads

    barrier = D.barrier       # rendezvous of the ranks + torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    launches0 = ext.launch_count()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    assert ext.launch_count() >= launches0 + 2 * args.steps, "HIP path did not run"
    elapsed = D.max_over_ranks(elapsed)

    # Per-kernel durations with HIP events on the stream the kernels are launched on
    # (torch's current stream), separate from the wall-clock region above.
    fwd_ms, bwd_ms = [], []
    for _ in range(args.steps):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        MSDA.ms_deform_attn_forward(v, ss, lsi, loc, aw, 128)
        e1.record()
        MSDA.ms_deform_attn_backward(v, ss, lsi, loc, aw, go, 128)
        e2.record()
        e2.synchronize()
        fwd_ms.append(e0.elapsed_time(e1))
        bwd_ms.append(e1.elapsed_time(e2))
    fwd_avg = sum(fwd_ms) / len(fwd_ms)
    bwd_avg = sum(bwd_ms) / len(bwd_ms)

    if rank != 0:
        D.shutdown()
        return

    ms_per_step = elapsed / args.steps * 1e3
    total_samples = samples * world
    value = total_samples / (elapsed / args.steps) / 1e9
    if args.dtype == "bf16":      # grad_value is returned in fp32: 22.67 + 45.33 = 68.00 B/sample
        fwd_b = synthetic.algorithmic_bytes_per_sample(value_bytes=2)[0]
        bwd_b = (32 * 2 + 32 * 2 + 32 * 4) / 12 + 24
    else:
        fwd_b, bwd_b = synthetic.algorithmic_bytes_per_sample()
    # dominant kernel = the backward call: the binning passes (zero, count, scan, scatter) and
    # msda_bwd_binned_kernel (~89 % of it) on the same stream; all are inside the event bracket
    bwd_gbs = samples * bwd_b / (bwd_avg * 1e-3) / 1e9
    fwd_gbs = samples * fwd_b / (fwd_avg * 1e-3) / 1e9
    both_gbs = samples * (fwd_b + bwd_b) / ((fwd_avg + bwd_avg) * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("%s/%s" % (args.workload, args.dist), {}).get(
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
        "config": {"workload": "%s: MSDeformAttn fwd+bwd, bs=%d/GPU, %d^2 input, levels %s, "
                               "M=8, D=32, P=4, Lq=S=%d, %s locations"
                               % (args.workload, batch, image,
                                  "/".join("%dx%d" % s for s in shapes),
                                  sum(h * w for h, w in shapes), args.dist),
                   "samples_per_step_per_gpu": samples, "parallelism": "batch-sharded x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "msda_backward call = msda_bwd_bin_{zero,count,scan,scatter} + msda_bwd_binned_kernel",
                     "achieved": round(bwd_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(bwd_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(samples * bwd_b),
                     "avg_launch_ms": round(bwd_avg, 4)},
        "detail": {"fwd_ms": round(fwd_avg, 4), "bwd_ms": round(bwd_avg, 4),
                   "fwd_gsamples_s": round(samples / fwd_avg / 1e6, 3),
                   "bwd_gsamples_s": round(samples / bwd_avg / 1e6, 3),
                   "fwd_hbm_frac": round(fwd_gbs / HBM_PEAK_GBS, 5),
                   "fwd_bwd_hbm_frac": round(both_gbs / HBM_PEAK_GBS, 5),
                   "fwd_bwd_gsamples_s_events": round(samples / (fwd_avg + bwd_avg) / 1e6, 3)},
    }
    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(shapes, args.dist, args.cpu_images, args.cpu_reps)
    print(json.dumps(result), flush=True)
    D.shutdown()


if __name__ == "__main__":
    main()
