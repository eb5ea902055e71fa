"""time the backward through the binding with entry regions of s entries per quad (whole batch in one group)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from instaorder_amd import ext, synthetic
import MultiScaleDeformableAttention as MSDA
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
image = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
inp = synthetic.make_inputs(batch, synthetic.encoder_shapes(image), dist="model", seed=0, device="cuda")
v = inp["value"]
ba = (v, inp["spatial_shapes"], inp["level_start_index"], inp["sampling_locations"], inp["attention_weights"], inp["grad_output"], 128)
n, s, m, d = v.shape
lq = inp["sampling_locations"].shape[1]; L = 3
Q = n * lq * m * L
fixed = 2 * (n * m * s + 2) + 64 + Q
ext.workspace_mode = "full"
for rep in range(3):
    for slots in (16, 8, 6, 5, 4, 3):
        ext.workspace_limit_bytes = None if slots == 16 else 4 * (fixed + slots * Q + 4 * lq) + 256
        for _ in range(5): MSDA.ms_deform_attn_backward(*ba)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): MSDA.ms_deform_attn_backward(*ba)
        e1.record(); e1.synchronize()
        print("bs%d %d^2 slots %2d (%.2f GB) bwd %.3f ms" % (batch, image, slots, 4 * (fixed + slots * Q) / 1e9, e0.elapsed_time(e1) / 20), flush=True)
