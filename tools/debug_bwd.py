"""Diagnostic: compare the HIP backward with the C oracle on one synthetic case and print the
samples whose grad_sampling_loc / grad_attn_weight disagree (GPU box; uses oracle/ as checker)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instaorder_amd import synthetic
from oracle import msda_oracle
import MultiScaleDeformableAttention as MSDA

shapes = [(40, 23), (17, 33), (5, 70)]
inp = synthetic.make_inputs(2, shapes, heads=8, channels=32, points=4, dist="uniform", seed=21, num_query=300)
inp["sampling_locations"] = inp["sampling_locations"] * 1.2 - 0.1
dev = {k: v.cuda() for k, v in inp.items()}
gv, gl, ga = MSDA.ms_deform_attn_backward(dev["value"], dev["spatial_shapes"], dev["level_start_index"],
                                          dev["sampling_locations"], dev["attention_weights"], dev["grad_output"], 128)
n = lambda k: inp[k].double().numpy() if inp[k].is_floating_point() else inp[k].numpy()
o_gv, o_gl, o_ga = msda_oracle.backward(n("value"), n("spatial_shapes"), n("level_start_index"),
                                        n("sampling_locations"), n("attention_weights"), n("grad_output"))
gl = gl.cpu().numpy(); ga = ga.cpu().numpy()
err = np.abs(gl - o_gl).max(-1)
scale = np.abs(o_gl).max()
bad = np.argwhere(err > 1e-4 * scale)
print("bad samples:", len(bad), "of", err.size, " grad_aw max err", np.abs(ga - o_ga).max())
loc = inp["sampling_locations"].numpy()
for b in bad[:25]:
    nn, q, m, l, p = b
    H, W = shapes[l]
    x = loc[nn, q, m, l, p, 0] * W - 0.5; y = loc[nn, q, m, l, p, 1] * H - 0.5
    print(tuple(int(v) for v in b), "x=%.3f y=%.3f (W=%d H=%d)" % (x, y, W, H), "got", gl[nn, q, m, l, p], "want", o_gl[nn, q, m, l, p],
          "ga", ga[nn, q, m, l, p], o_ga[nn, q, m, l, p])
