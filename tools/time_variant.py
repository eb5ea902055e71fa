"""Diagnostic: time the backward of C2 with an alternative build of the library
(python tools/time_variant.py tools/bin/<lib>.so); results of experiment builds may be wrong."""
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    MSDA.ms_deform_attn_backward(*args)
e1.record()
e1.synchronize()
print("%s: backward %.3f ms" % (ext._LIB_PATH, e0.elapsed_time(e1) / 20))
