"""Diagnostic: per-kernel device time of one backward call of C2 with an alternative build
(python tools/bwd_kernel_times.py [lib.so]); uses torch.profiler (kineto/roctracer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instaorder_amd import ext, synthetic
if len(sys.argv) > 1:
    ext._LIB_PATH = os.path.abspath(sys.argv[1])
import MultiScaleDeformableAttention as MSDA
inp = synthetic.make_inputs(16, synthetic.encoder_shapes(1024), dist="model", seed=0, device="cuda")
args = (inp["value"], inp["spatial_shapes"], inp["level_start_index"], inp["sampling_locations"],
        inp["attention_weights"], inp["grad_output"], 128)
for _ in range(5):
    MSDA.ms_deform_attn_backward(*args)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        MSDA.ms_deform_attn_backward(*args)
    torch.cuda.synchronize()
print(ext._LIB_PATH)
for e in prof.key_averages():
    if "msda" in e.key:
        print("  %-60s %8.1f us" % (e.key.split("(")[0][-60:], e.device_time_total / e.count))
