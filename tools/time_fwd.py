"""Diagnostic: time the forward (and backward) of a workload with the library as built
(python tools/time_fwd.py [workload] [dtype]); MSDA_FWD_IMPL / MSDA_BWD_IMPL select variants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instaorder_amd import ext, synthetic
import MultiScaleDeformableAttention as MSDA
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
batch, image = {"C2": (16, 1024), "C3": (32, 1024), "C5": (2, 1536), "C5x16": (16, 1536)}[wl]
dist = sys.argv[3] if len(sys.argv) > 3 else "model"
inp = synthetic.make_inputs(batch, synthetic.encoder_shapes(image), dist=dist, seed=0, device="cuda")
v, go = inp["value"], inp["grad_output"]
if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    v, go = v.bfloat16(), go.bfloat16()
a = (v, inp["spatial_shapes"], inp["level_start_index"], inp["sampling_locations"], inp["attention_weights"])
for _ in range(5):
    out = MSDA.ms_deform_attn_forward(*a, 128); MSDA.ms_deform_attn_backward(*a, go, 128)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 30
ev[0].record()
for _ in range(n): MSDA.ms_deform_attn_forward(*a, 128)
ev[1].record()
for _ in range(n): MSDA.ms_deform_attn_backward(*a, go, 128)
ev[2].record(); ev[2].synchronize()
print("%s %s %s fwd_impl=%s: fwd %.4f ms  bwd %.4f ms  out checksum %.6f" % (
    wl, dist, v.dtype, os.environ.get("MSDA_FWD_IMPL", "default"), ev[0].elapsed_time(ev[1]) / n,
    ev[1].elapsed_time(ev[2]) / n, out.double().abs().sum().item()))
