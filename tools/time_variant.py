"""Diagnostic: time forward and backward with an alternative build of the library
(python tools/time_variant.py tools/bin/<lib>.so [batch] [image] [dist] [dtype]); results of
experiment builds may be wrong -- this measures, it does not check."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instaorder_amd import ext, synthetic
if len(sys.argv) > 1 and sys.argv[1] != "-":
    ext._LIB_PATH = os.path.abspath(sys.argv[1])
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
image = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dist = sys.argv[4] if len(sys.argv) > 4 else "model"
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
import MultiScaleDeformableAttention as MSDA
inp = synthetic.make_inputs(batch, synthetic.encoder_shapes(image), dist=dist, seed=0, device="cuda")
v, go = inp["value"], inp["grad_output"]
if dtype == "bf16":
    v, go = v.bfloat16(), go.bfloat16()
fa = (v, inp["spatial_shapes"], inp["level_start_index"], inp["sampling_locations"], inp["attention_weights"], 128)
ba = fa[:5] + (go, 128)
def timed(fn, args, reps=20):
    for _ in range(5):
        fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(*args)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps
f = timed(MSDA.ms_deform_attn_forward, fa)
b = timed(MSDA.ms_deform_attn_backward, ba)
print("%-40s bs%d %d^2 %s %s: fwd %.3f ms  bwd %.3f ms" % (os.path.basename(ext._LIB_PATH), batch, image, dist, dtype, f, b), flush=True)
