"""ctypes front-end of the C oracle (oracle/msda_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path never imports this module.

All functions take and return numpy arrays laid out exactly as the reference
op's tensors (mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_attn_cuda.cu:25-85):
value [N,S,M,D], spatial_shapes int64 [L,2] (H,W), level_start_index int64 [L],
sampling_locations [N,Lq,M,L,P,2] (x,y), attention_weights [N,Lq,M,L,P],
output / grad_output [N,Lq,M*D].
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "msda_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmsda_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lsi, loc, aw):
    dt = value.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError("oracle supports float32/float64, got %s" % dt)
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    aw = np.ascontiguousarray(aw, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and shapes.shape == (L, 2) and lsi.shape == (L,)
    assert aw.shape == (N, Lq, M, L, P)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    return value, shapes, lsi, loc, aw, (N, S, M, D, L, Lq, P)


def forward(value, shapes, lsi, loc, aw):
    value, shapes, lsi, loc, aw, dims = _prep(value, shapes, lsi, loc, aw)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=value.dtype)
    fn = getattr(_load(), "msda_oracle_forward_" + ("f32" if value.dtype == np.float32 else "f64"))
    fn.restype = None
    fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(aw),
       *[ctypes.c_int(x) for x in (N, S, M, D, L, Lq, P)], _ptr(out))
    return out


def backward(value, shapes, lsi, loc, aw, grad_out):
    value, shapes, lsi, loc, aw, dims = _prep(value, shapes, lsi, loc, aw)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype)
    assert grad_out.shape == (N, Lq, M * D)
    gv = np.empty_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(aw)
    fn = getattr(_load(), "msda_oracle_backward_" + ("f32" if value.dtype == np.float32 else "f64"))
    fn.restype = None
    fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(aw), _ptr(grad_out),
       *[ctypes.c_int(x) for x in (N, S, M, D, L, Lq, P)], _ptr(gv), _ptr(gl), _ptr(ga))
    return gv, gl, ga
