import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import MultiScaleDeformableAttention as MSDA
from instaorder_amd import synthetic
from oracle import msda_oracle
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
img = int(sys.argv[2]) if len(sys.argv) > 2 else 256
inp = synthetic.make_inputs(3, synthetic.encoder_shapes(img), dist=dist, seed=11)
dev = {k: v.cuda() for k, v in inp.items()}
gv, gl, ga = MSDA.ms_deform_attn_backward(dev["value"], dev["spatial_shapes"], dev["level_start_index"], dev["sampling_locations"], dev["attention_weights"], dev["grad_output"], 128)
n = lambda k: inp[k].double().numpy() if inp[k].is_floating_point() else inp[k].numpy()
o_gv, o_gl, o_ga = msda_oracle.backward(n("value"), n("spatial_shapes"), n("level_start_index"), n("sampling_locations"), n("attention_weights"), n("grad_output"))
gl = gl.cpu().double().numpy(); ga = ga.cpu().double().numpy(); gv = gv.cpu().double().numpy()
print("gv err", np.abs(gv-o_gv).max()/np.abs(o_gv).max())
print("ga err", np.abs(ga-o_ga).max()/np.abs(o_ga).max())
print("gl err", np.abs(gl-o_gl).max()/np.abs(o_gl).max())
bad = np.argwhere(np.abs(gl-o_gl) > 1e-3*np.abs(o_gl).max())
print("bad count", len(bad), "of", gl.size)
shapes = n("spatial_shapes")
loc = n("sampling_locations")
for b in bad[:20]:
    nn,q,m,l,p,xy = b
    H,W = shapes[l]
    x = loc[nn,q,m,l,p,0]*W-0.5; y = loc[nn,q,m,l,p,1]*H-0.5
    print(b, "got", gl[tuple(b)], "want", o_gl[tuple(b)], "pix", x, y, "HW", H, W, "ga got/want", ga[nn,q,m,l,p], o_ga[nn,q,m,l,p])
