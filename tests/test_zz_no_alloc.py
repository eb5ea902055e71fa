"""GPU: msda_backward_f32_noalloc -- the one entry point that allocates nothing (its sort workspace
is a device array built into the library) -- called through the C ABI and checked against
msda_backward_f32_ws (caller-provided workspace) on the same inputs: two calls in a row, the second
on a smaller input, so whatever the first call left in the shared scratch array must not leak into
the second.  Tolerance: the order of a tile's list, hence of the fp32 sums, is not fixed from run to
run (tests/test_gpu_parity.py::test_backward_run_to_run_agreement), so "equal" means equal to
rounding; against the north-star's 1e-4 gradient bar this is 10x tighter.

Collected last (file name) so a failure here cannot hide another test behind `pytest -x`."""
import ctypes

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

ORDER_TOL = 1e-5


def _call(lib, fn, inp, *extra):
    v, ss, lsi = inp["value"], inp["spatial_shapes"], inp["level_start_index"]
    loc, aw, go = inp["sampling_locations"], inp["attention_weights"], inp["grad_output"]
    n, s, m, d = v.shape
    nl, lq, p = ss.shape[0], loc.shape[1], loc.shape[4]
    gv, gl, ga = torch.full_like(v, 7.0), torch.full_like(loc, 7.0), torch.full_like(aw, 7.0)
    rc = fn(v.data_ptr(), ss.data_ptr(), lsi.data_ptr(), loc.data_ptr(), aw.data_ptr(), go.data_ptr(),
            n, s, m, d, nl, lq, p, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), *extra,
            torch.cuda.current_stream().cuda_stream)
    return rc, gv, gl, ga


def _dims(inp):
    v, ss, loc = inp["value"], inp["spatial_shapes"], inp["sampling_locations"]
    n, s, m, d = v.shape
    return n, s, m, d, ss.shape[0], loc.shape[1], loc.shape[4]


def test_no_alloc_backward_matches_workspace_backward():  # noqa: C901
    assert torch.cuda.is_available(), "needs the MI355X"
    from instaorder_amd import ext, synthetic
    lib = ext._load()
    I, P = ctypes.c_int, ctypes.c_void_p
    lib.msda_backward_f32_noalloc.argtypes = [P] * 6 + [I] * 7 + [P] * 4
    lib.msda_backward_f32_noalloc.restype = I
    lib.msda_backward_noalloc_max_workspace_bytes.restype = ctypes.c_size_t
    cap = int(lib.msda_backward_noalloc_max_workspace_bytes())
    assert cap == 64 << 20

    big = synthetic.make_inputs(2, synthetic.encoder_shapes(512), dist="model", seed=31, device="cuda")
    small = synthetic.make_inputs(1, [(5, 9), (20, 13)], heads=3, channels=32, points=2, dist="uniform",
                                  seed=32, num_query=41, device="cuda")
    generic = synthetic.make_inputs(1, [(6, 4), (3, 2)], heads=2, channels=30, points=2, dist="uniform",
                                    seed=33, num_query=9, device="cuda")       # D = 30: no workspace at all
    for inp in (big, small, big, generic):
        need = int(lib.msda_backward_workspace_bytes(*_dims(inp)))
        assert need <= cap
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
        rc, gv0, gl0, ga0 = _call(lib, lib.msda_backward_f32_ws, inp, ws.data_ptr(), need)
        assert rc == 0, lib.msda_last_error()
        torch.cuda.synchronize()
        # no allocation between here and the synchronize: outputs exist already
        before = ext.launch_count()
        rc, gv, gl, ga = _call(lib, lib.msda_backward_f32_noalloc, inp)
        assert rc == 0, lib.msda_last_error()
        torch.cuda.synchronize()
        assert ext.launch_count() > before
        for got, want in ((gv, gv0), (gl, gl0), (ga, ga0)):
            assert torch.isfinite(got).all()
            assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < ORDER_TOL

    # a batch whose lists exceed the built-in array goes through it in groups of images: same results
    two = synthetic.make_inputs(2, synthetic.encoder_shapes(1024), dist="model", seed=34, device="cuda")
    need = int(lib.msda_backward_workspace_bytes(*_dims(two)))
    assert need > cap
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    rc, gv0, gl0, ga0 = _call(lib, lib.msda_backward_f32_ws, two, ws.data_ptr(), need)
    assert rc == 0, lib.msda_last_error()
    rc, gv, gl, ga = _call(lib, lib.msda_backward_f32_noalloc, two)
    assert rc == 0, lib.msda_last_error()
    torch.cuda.synchronize()
    for got, want in ((gv, gv0), (gl, gl0), (ga, ga0)):
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < ORDER_TOL
    del two, ws, gv0, gl0, ga0, gv, gl, ga

    # one image whose worst case (70 MB at 1536^2) exceeds the array: taken with a smaller entry region, in rounds
    lib.msda_backward_min_workspace_bytes.restype = ctypes.c_size_t
    large = synthetic.make_inputs(1, synthetic.encoder_shapes(1536), dist="model", seed=35, device="cuda")
    need = int(lib.msda_backward_workspace_bytes(*_dims(large)))
    assert int(lib.msda_backward_min_workspace_bytes(*_dims(large))) < cap < need
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    rc, gv0, gl0, ga0 = _call(lib, lib.msda_backward_f32_ws, large, ws.data_ptr(), need)
    assert rc == 0, lib.msda_last_error()
    rc, gv, gl, ga = _call(lib, lib.msda_backward_f32_noalloc, large)
    assert rc == 0, lib.msda_last_error()
    torch.cuda.synchronize()
    for got, want in ((gv, gv0), (gl, gl0), (ga, ga0)):
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < ORDER_TOL
    del large, ws, gv0, gl0, ga0, gv, gl, ga

    # a problem whose ONE image cannot run in the array even at one entry per quad is refused, outputs untouched
    huge = synthetic.make_inputs(1, synthetic.encoder_shapes(3584), dist="uniform", seed=35, device="cuda")
    assert int(lib.msda_backward_min_workspace_bytes(*_dims(huge))) > cap
    rc, gv, gl, ga = _call(lib, lib.msda_backward_f32_noalloc, huge)
    assert rc == 1 and b"built-in scratch" in lib.msda_last_error()
    torch.cuda.synchronize()
    assert float((gv - 7.0).abs().max()) == 0.0 and float((ga - 7.0).abs().max()) == 0.0


def test_no_alloc_backward_allocates_nothing_and_matches_the_oracle(oracle):
    """What the entry point is for: no device memory is taken between its call and its completion (HIP's own
    free-memory figure and torch's allocator statistics stay put), it can be captured into a HIP graph -- first
    call on the device inside the capture included, in a fresh process -- and replayed, and its results are the
    oracle's."""
    import os
    import subprocess
    import sys
    from instaorder_amd import ext, synthetic
    lib = ext._load()
    I, P = ctypes.c_int, ctypes.c_void_p
    lib.msda_backward_f32_noalloc.argtypes = [P] * 6 + [I] * 7 + [P] * 4
    lib.msda_backward_f32_noalloc.restype = I
    inp = synthetic.make_inputs(2, synthetic.encoder_shapes(256), dist="model", seed=36)
    n = lambda k: inp[k].double().numpy() if inp[k].is_floating_point() else inp[k].numpy()
    o_gv, o_gl, o_ga = oracle.backward(n("value"), n("spatial_shapes"), n("level_start_index"),
                                       n("sampling_locations"), n("attention_weights"), n("grad_output"))
    dev = {k: t.cuda() for k, t in inp.items()}
    _call(lib, lib.msda_backward_f32_noalloc, dev)           # first call on the device: the symbol look-up
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    stats0 = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
    v, ss, lsi = dev["value"], dev["spatial_shapes"], dev["level_start_index"]
    loc, aw, go = dev["sampling_locations"], dev["attention_weights"], dev["grad_output"]
    gv, gl, ga = torch.empty_like(v), torch.empty_like(loc), torch.empty_like(aw)
    stats1 = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
    free1 = torch.cuda.mem_get_info()[0]
    nb, s, m, d = v.shape
    rc = lib.msda_backward_f32_noalloc(v.data_ptr(), ss.data_ptr(), lsi.data_ptr(), loc.data_ptr(), aw.data_ptr(),
                                       go.data_ptr(), nb, s, m, d, ss.shape[0], loc.shape[1], loc.shape[4],
                                       gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.msda_last_error()
    torch.cuda.synchronize()
    assert (torch.cuda.memory_allocated(), torch.cuda.memory_reserved()) == stats1
    assert torch.cuda.mem_get_info()[0] == free1, "device memory was taken during the call"
    assert free1 <= free0 and stats1[0] >= stats0[0]
    assert rel_err(gv.cpu().numpy(), o_gv) < 1e-4
    assert rel_err(gl.cpu().numpy(), o_gl) < 1e-4
    assert rel_err(ga.cpu().numpy(), o_ga) < 1e-4

    # graph capture in a fresh process, so that the first call on the device happens under capture
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import ctypes, sys
sys.path.insert(0, %r)
import torch
from instaorder_amd import ext, synthetic
lib = ext._load()
I, P = ctypes.c_int, ctypes.c_void_p
lib.msda_backward_f32_noalloc.argtypes = [P] * 6 + [I] * 7 + [P] * 4
lib.msda_backward_f32_noalloc.restype = I
lib.msda_backward_f32_ws.argtypes = [P] * 6 + [I] * 7 + [P] * 4 + [ctypes.c_size_t, P]
inp = synthetic.make_inputs(2, synthetic.encoder_shapes(256), dist="model", seed=36, device="cuda")
v, ss, lsi = inp["value"], inp["spatial_shapes"], inp["level_start_index"]
loc, aw, go = inp["sampling_locations"], inp["attention_weights"], inp["grad_output"]
n, s, m, d = v.shape
dims = (n, s, m, d, ss.shape[0], loc.shape[1], loc.shape[4])
out = [torch.zeros_like(v), torch.zeros_like(loc), torch.zeros_like(aw)]
ptrs = (v.data_ptr(), ss.data_ptr(), lsi.data_ptr(), loc.data_ptr(), aw.data_ptr(), go.data_ptr())
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        rc = lib.msda_backward_f32_noalloc(*ptrs, *dims, *(t.data_ptr() for t in out), side.cuda_stream)
        assert rc == 0, lib.msda_last_error()
torch.cuda.synchronize()
assert float(out[0].abs().max()) == 0.0          # captured, not run
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
need = int(lib.msda_backward_workspace_bytes(*dims))
ws = torch.empty(need, dtype=torch.uint8, device="cuda")
ref = [torch.zeros_like(v), torch.zeros_like(loc), torch.zeros_like(aw)]
rc = lib.msda_backward_f32_ws(*ptrs, *dims, *(t.data_ptr() for t in ref), ws.data_ptr(), need,
                              torch.cuda.current_stream().cuda_stream)
assert rc == 0
torch.cuda.synchronize()
for a, b in zip(out, ref):
    err = float((a - b).abs().max() / b.abs().max())
    assert err < 1e-5, err
print("captured-ok")
""" % root
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110)
    assert run.returncode == 0 and "captured-ok" in run.stdout, run.stderr[-3000:]
