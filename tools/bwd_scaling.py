"""Per-image backward/forward time vs batch size (is the backward memory-bound?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import MultiScaleDeformableAttention as MSDA
from instaorder_amd import synthetic
for batch in (2, 4, 8, 16, 32):
    inp = synthetic.make_inputs(batch, synthetic.encoder_shapes(1024), dist="model", seed=0, device="cuda")
    a = (inp["value"], inp["spatial_shapes"], inp["level_start_index"], inp["sampling_locations"], inp["attention_weights"])
    go = inp["grad_output"]
    for _ in range(3):
        MSDA.ms_deform_attn_forward(*a, 128); MSDA.ms_deform_attn_backward(*a, go, 128)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 10
    ev[0].record()
    for _ in range(n): MSDA.ms_deform_attn_forward(*a, 128)
    ev[1].record()
    for _ in range(n): MSDA.ms_deform_attn_backward(*a, go, 128)
    ev[2].record(); ev[2].synchronize()
    f = ev[0].elapsed_time(ev[1]) / n; b = ev[1].elapsed_time(ev[2]) / n
    print("batch %2d  fwd %.3f ms (%.4f /img)  bwd %.3f ms (%.4f /img)" % (batch, f, f / batch, b, b / batch))
