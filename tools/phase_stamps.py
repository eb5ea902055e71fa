"""Diagnostic: per-phase cycle shares of the backward tile kernel (needs tools/bin/libmsda_hip_stamps.so,
built with -DMSDA_PHASE_STAMPS).  Shares only -- the stamped build is slower than the real one."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instaorder_amd import ext, synthetic
ext._LIB_PATH = os.path.join(ROOT, "tools", "bin", "libmsda_hip_stamps.so")
import MultiScaleDeformableAttention as MSDA
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
inp = synthetic.make_inputs(batch, synthetic.encoder_shapes(1024), dist="model", seed=0, device="cuda")
args = (inp["value"], inp["spatial_shapes"], inp["level_start_index"], inp["sampling_locations"], inp["attention_weights"], inp["grad_output"], 128)
for _ in range(3): MSDA.ms_deform_attn_backward(*args)
lib = ext._load()
buf = (ctypes.c_ulonglong * 16)()
lib.msda_debug_phase_cycles(buf)
for _ in range(5): MSDA.ms_deform_attn_backward(*args)
lib.msda_debug_phase_cycles(buf)
names = ["item setup (decode, peek)", "A: prefetch issue + max", "barrier 1", "C: rows->LDS, home push, zero, row prefetch", "C: scatter", "D accumulate", "E home grads", "barrier 2", "A: wait prefetched sample", "A: geometry", "A: atomics issue", "item flush (stores issued)"]
tot = sum(buf[i] for i in range(12))
for i, nm in enumerate(names):
    print("%-30s %6.2f %%   %.3e cycles" % (nm, 100.0 * buf[i] / tot, buf[i]))
