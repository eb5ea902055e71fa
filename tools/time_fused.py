"""Diagnostic: time the fused boundary's forward and backward (python tools/time_fused.py [lib.so|-] [batch] [image] [dtype])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instaorder_amd import ext, synthetic
if len(sys.argv) > 1 and sys.argv[1] != "-":
    ext._LIB_PATH = os.path.abspath(sys.argv[1])
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
image = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
shapes = synthetic.encoder_shapes(image)
inp = synthetic.make_inputs(batch, shapes, dist="model", seed=0, device="cuda")
v, go = inp["value"], inp["grad_output"]
if dtype == "bf16":
    v, go = v.bfloat16(), go.bfloat16()
ss, lsi, loc = inp["spatial_shapes"], inp["level_start_index"], inp["sampling_locations"]
n, lq, m, nl, p, _ = loc.shape
ref = synthetic.reference_points(shapes).cuda()[None, :, None, :].expand(n, lq, nl, 2).contiguous()
wh = torch.tensor([(w, h) for h, w in shapes], dtype=torch.float32, device="cuda")
off = ((loc - ref[:, :, None, :, None, :]) * wh[None, None, None, :, None, :]).contiguous()
logits = torch.log(inp["attention_weights"].clamp_min(1e-20)).reshape(n, lq, m, nl * p).contiguous()
fa = (v, ss, lsi, ref, off, logits)
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
f = timed(ext.ms_deform_attn_fused_forward, fa)
b = timed(ext.ms_deform_attn_fused_backward, fa + (go,))
print("%-28s fused bs%d %d^2 %s: fwd %.3f ms  bwd %.3f ms" % (os.path.basename(ext._LIB_PATH), batch, image, dtype, f, b), flush=True)
