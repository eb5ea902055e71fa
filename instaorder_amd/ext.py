"""Host side of the MSDA extension boundary: the two functions the reference's pybind
module ``MultiScaleDeformableAttention`` exports (reference
mask2former/modeling/pixel_decoder/ops/src/vision.cpp:18-21), implemented on top of
the C ABI of libmsda_hip.so (include/msda_hip.h).

Mirrors, check for check, ``ms_deform_attn_forward/backward`` (ops/src/ms_deform_attn.h:25-66)
and ``ms_deform_attn_cuda_forward/backward`` (ops/src/cuda/ms_deform_attn_cuda.cu:25-158):
same positional signatures, same preconditions, same error type (RuntimeError), outputs
allocated here on ``value``'s device and dtype, kernels enqueued on torch's current HIP
stream without synchronising.  There is NO fallback: a missing library, a CPU tensor or a
failed launch raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsda_hip.so")
_lib = None

_I = ctypes.c_int
_P = ctypes.c_void_p


def _load():
    """dlopen libmsda_hip.so (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            "libmsda_hip.so is not built: run `python -m instaorder_amd.build_ext` "
            "(needs hipcc); there is no fallback implementation")
    lib = ctypes.CDLL(_LIB_PATH)
    fwd_args = [_P] * 5 + [_I] * 7 + [_P, _P]
    bwd_args = [_P] * 6 + [_I] * 7 + [_P] * 4
    for suffix in ("f32", "f64"):
        f = getattr(lib, "msda_forward_" + suffix)
        f.argtypes, f.restype = fwd_args, _I
        b = getattr(lib, "msda_backward_" + suffix)
        b.argtypes, b.restype = bwd_args, _I
    lib.msda_forward_bf16.argtypes, lib.msda_forward_bf16.restype = fwd_args, _I
    lib.msda_backward_bf16.argtypes, lib.msda_backward_bf16.restype = bwd_args, _I
    ws_args = [_P] * 6 + [_I] * 7 + [_P] * 4 + [ctypes.c_size_t, _P]
    for suffix in ("f32", "bf16"):
        b = getattr(lib, "msda_backward_%s_ws" % suffix)
        b.argtypes, b.restype = ws_args, _I
    for q in ("msda_backward_workspace_bytes", "msda_backward_min_workspace_bytes",
              "msda_backward_bf16_gv16_workspace_bytes"):
        getattr(lib, q).argtypes = [_I] * 7
        getattr(lib, q).restype = ctypes.c_size_t
    lib.msda_backward_compact_workspace_bytes.argtypes = [_I] * 8
    lib.msda_backward_compact_workspace_bytes.restype = ctypes.c_size_t
    lib.msda_backward_bf16_gv16_ws.argtypes, lib.msda_backward_bf16_gv16_ws.restype = ws_args, _I
    lib.msda_last_error.restype = ctypes.c_char_p
    lib.msda_launch_count.restype = ctypes.c_uint64
    lib.msda_abi_version.restype = _I
    for suffix in ("f32", "bf16"):
        f = getattr(lib, "msda_fused_forward_" + suffix)
        f.argtypes, f.restype = [_P] * 6 + [_I] * 7 + [_P, _P], _I
        b = getattr(lib, "msda_fused_backward_" + suffix)
        b.argtypes, b.restype = [_P] * 7 + [_I] * 7 + [_P] * 4 + [ctypes.c_size_t, _P], _I
    lib.msda_fused_backward_workspace_bytes.argtypes = [_I] * 7
    lib.msda_fused_backward_workspace_bytes.restype = ctypes.c_size_t
    if lib.msda_abi_version() != 5:
        raise RuntimeError("libmsda_hip.so ABI version mismatch")
    _lib = lib
    return lib


# Upper limit (bytes) on the backward's sort workspace, or None: the lists of the whole batch at their worst case
# (the fastest: one pass).  With a limit the library takes the lists through a smaller entry region in rounds and
# the batch in groups of images (include/msda_hip.h); never less than the minimum for one image is allocated.
workspace_limit_bytes = None

# How much the binding allocates when no limit is set: "full" = the worst case of sixteen list entries per
# (query, head, level): one round whatever the locations, the fastest; or "compact" = four
# (msda_backward_compact_workspace_bytes: four rounds, of which model-like locations use the first; 0.37 GB instead
# of 1.17 GB at bs 32 / 1024^2 for +0.4 % of the backward, +2 % at bs 2 / 1536^2).
workspace_mode = "full"


def _workspace_bytes(lib, dims, bf16):
    if workspace_mode == "full":
        query = lib.msda_backward_bf16_gv16_workspace_bytes if bf16 else lib.msda_backward_workspace_bytes
        nbytes = int(query(*dims))
    elif workspace_mode == "compact":
        nbytes = int(lib.msda_backward_compact_workspace_bytes(*dims, 1 if bf16 else 0))
    else:
        raise ValueError("instaorder_amd.ext.workspace_mode must be 'compact' or 'full'")
    if workspace_limit_bytes is None or nbytes <= workspace_limit_bytes:
        return nbytes
    return max(int(workspace_limit_bytes), int(lib.msda_backward_min_workspace_bytes(*dims)))


def launch_count():
    """Kernel launches issued through the library so far (proof the HIP path ran)."""
    return int(_load().msda_launch_count())


_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}


def _check_inputs(named):
    # ms_deform_attn_cuda.cu:33-43 / :98-110 -- contiguity first, then device
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
    for name, t in named:
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)


def _common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if value.dtype not in _SUFFIX:
        # AT_DISPATCH_FLOATING_TYPES (ms_deform_attn_cuda.cu:69): float and double only
        raise RuntimeError('"ms_deform_attn" not implemented for \'%s\'' % value.dtype)
    # bf16 variant (an extension, not reference behaviour): bf16 value, fp32 locations/weights
    side = torch.float32 if value.dtype == torch.bfloat16 else value.dtype
    if sampling_loc.dtype != side or attn_weight.dtype != side:
        raise RuntimeError("expected sampling_loc and attn_weight of dtype %s for a %s value"
                           % (side, value.dtype))
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 tensors")
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("expected value [N,S,M,D], sampling_loc [N,Lq,M,L,P,2], "
                           "attn_weight [N,Lq,M,L,P]")
    batch, spatial_size, num_heads, channels = value.shape
    num_levels = spatial_shapes.shape[0]
    num_query, num_point = sampling_loc.shape[1], sampling_loc.shape[4]
    if (tuple(sampling_loc.shape) != (batch, num_query, num_heads, num_levels, num_point, 2)
            or tuple(attn_weight.shape) != (batch, num_query, num_heads, num_levels, num_point)
            or tuple(spatial_shapes.shape) != (num_levels, 2)
            or tuple(level_start_index.shape) != (num_levels,)):
        raise RuntimeError("inconsistent MSDA argument shapes")
    step = min(batch, int(im2col_step))
    if batch > 0 and (step <= 0 or batch % step != 0):
        # ms_deform_attn_cuda.cu:55-57
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (batch, step))
    return batch, spatial_size, num_heads, channels, num_levels, num_query, num_point


def _raise(lib, what):
    raise RuntimeError("%s: %s" % (what, lib.msda_last_error().decode("utf-8", "replace")))


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step):
    """-> output [N, Lq, M*D].  Reference: ops/src/ms_deform_attn.h:25-44."""
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")        # ms_deform_attn.h:43
    _check_inputs((("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight)))
    n, s, m, d, nl, lq, npt = _common(value, spatial_shapes, level_start_index, sampling_loc,
                                      attn_weight, im2col_step)
    lib = _load()
    with torch.cuda.device(value.device):
        output = torch.empty((n, lq, m * d), dtype=value.dtype, device=value.device)
        stream = torch.cuda.current_stream().cuda_stream
        rc = getattr(lib, "msda_forward_" + _SUFFIX[value.dtype])(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), n, s, m, d, nl, lq, npt,
            output.data_ptr(), stream)
    if rc != 0:
        _raise(lib, "ms_deform_attn_forward")
    return output


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight].  Reference: ms_deform_attn.h:46-66.

    Level layouts the reference never builds (include/msda_hip.h): a level that is empty or leaves
    [0, spatial_size) gives zeros in all three gradients from every kernel; levels that SHARE rows give zeros
    from the channels == 32 kernels (which own whole rows) but are computed -- shared rows receive both levels'
    contributions, as in the reference -- by the generic kernels (other channel counts, fp64) and by every
    forward.  grad_value has value's dtype (a bf16 value: the fp32 sums rounded once)."""
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")        # ms_deform_attn.h:65
    _check_inputs((("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight), ("grad_output", grad_output)))
    n, s, m, d, nl, lq, npt = _common(value, spatial_shapes, level_start_index, sampling_loc,
                                      attn_weight, im2col_step)
    if grad_output.dtype != value.dtype or grad_output.numel() != n * lq * m * d:
        raise RuntimeError("grad_output must be [N, Lq, M*D] with value's dtype")
    lib = _load()
    with torch.cuda.device(value.device):
        # fully written by the library, in value's dtype (bf16 variant: the fp32 sums of a row are rounded
        # once, by the tile that owns the row -- msda_backward_bf16_gv16_ws, include/msda_hip.h)
        grad_value = torch.empty_like(value)
        grad_sampling_loc = torch.empty_like(sampling_loc)
        grad_attn_weight = torch.empty_like(attn_weight)
        stream = torch.cuda.current_stream().cuda_stream
        args = (value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                n, s, m, d, nl, lq, npt, grad_value.data_ptr(), grad_sampling_loc.data_ptr(),
                grad_attn_weight.data_ptr())
        if value.dtype == torch.float64:
            rc = lib.msda_backward_f64(*args, stream)
        else:
            # the sort-by-destination-tile workspace comes from torch's caching allocator
            # (the library's own entry points without _ws use the HIP stream-ordered pool)
            bf16 = value.dtype == torch.bfloat16
            nbytes = _workspace_bytes(lib, (n, s, m, d, nl, lq, npt), bf16)
            workspace = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=value.device)
            entry = lib.msda_backward_bf16_gv16_ws if bf16 else lib.msda_backward_f32_ws
            rc = entry(*args, workspace.data_ptr(), nbytes, stream)
    if rc != 0:
        _raise(lib, "ms_deform_attn_backward")
    return [grad_value, grad_sampling_loc, grad_attn_weight]


# ---------------------------------------------------------------------------------------------
# Fused boundary (opt-in): the op plus the softmax and the
# sampling-location arithmetic of its caller (reference ops/modules/ms_deform_attn.py:102-109).
# ---------------------------------------------------------------------------------------------
def fused_forward_supported(value, reference_points, num_levels, num_point, sampling_offsets=None,
                            attn_logits=None):
    """True when the fused entry points take the call (else use the unfused functions): every
    precondition of msda_fused_forward_* / msda_fused_backward_* and of _fused_common -- a GPU value
    of 32 channels per head in fp32 or bf16, fp32 reference points [N,Lq,L,2] / offsets / logits,
    num_levels * num_point <= 16, tensors of < 2^30 elements, a non-empty batch."""
    if not (value.is_cuda and value.dtype in (torch.float32, torch.bfloat16) and value.dim() == 4
            and value.shape[3] == 32 and reference_points.dim() == 4 and reference_points.shape[-1] == 2
            and reference_points.dtype == torch.float32 and num_levels * num_point <= 16):
        return False
    side = [t for t in (sampling_offsets, attn_logits) if t is not None]
    if any(t.dtype != torch.float32 or not t.is_cuda for t in side):
        return False
    limit = 1 << 30
    n, _, m, _ = value.shape
    lq = reference_points.shape[1]
    if value.numel() >= limit or n * lq * m * num_levels * num_point * 2 >= limit or n * lq * m * 32 >= limit:
        return False
    return n * lq > 0 and value.numel() > 0


def _fused_common(value, spatial_shapes, level_start_index, reference_points, sampling_offsets,
                  attn_logits):
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    named = (("value", value), ("spatial_shapes", spatial_shapes),
             ("level_start_index", level_start_index), ("reference_points", reference_points),
             ("sampling_offsets", sampling_offsets), ("attn_logits", attn_logits))
    _check_inputs(named)
    if value.dtype not in (torch.float32, torch.bfloat16) or any(
            t.dtype != torch.float32 for t in (reference_points, sampling_offsets, attn_logits)):
        raise RuntimeError("the fused MSDA entry points take an fp32 or bf16 value and fp32 reference "
                           "points, offsets and logits")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 tensors")
    if value.dim() != 4 or sampling_offsets.dim() != 6:
        raise RuntimeError("expected value [N,S,M,D], sampling_offsets [N,Lq,M,L,P,2]")
    n, s, m, d = value.shape
    nl = spatial_shapes.shape[0]
    lq, npt = sampling_offsets.shape[1], sampling_offsets.shape[4]
    if (tuple(sampling_offsets.shape) != (n, lq, m, nl, npt, 2)
            or attn_logits.numel() != n * lq * m * nl * npt
            or tuple(reference_points.shape) != (n, lq, nl, 2)
            or tuple(spatial_shapes.shape) != (nl, 2) or tuple(level_start_index.shape) != (nl,)):
        raise RuntimeError("inconsistent fused MSDA argument shapes (reference_points must be "
                           "[N,Lq,L,2])")
    return n, s, m, d, nl, lq, npt


def ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, reference_points,
                                 sampling_offsets, attn_logits):
    """-> output [N, Lq, M*D] from the raw offsets [N,Lq,M,L,P,2] and logits [N,Lq,M,L*P]."""
    n, s, m, d, nl, lq, npt = _fused_common(value, spatial_shapes, level_start_index,
                                            reference_points, sampling_offsets, attn_logits)
    lib = _load()
    with torch.cuda.device(value.device):
        output = torch.empty((n, lq, m * d), dtype=value.dtype, device=value.device)
        rc = getattr(lib, "msda_fused_forward_" + _SUFFIX[value.dtype])(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            reference_points.data_ptr(), sampling_offsets.data_ptr(), attn_logits.data_ptr(),
            n, s, m, d, nl, lq, npt, output.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        _raise(lib, "ms_deform_attn_fused_forward")
    return output


def ms_deform_attn_fused_backward(value, spatial_shapes, level_start_index, reference_points,
                                  sampling_offsets, attn_logits, grad_output):
    """-> [grad_value, grad_sampling_offsets, grad_attn_logits]."""
    n, s, m, d, nl, lq, npt = _fused_common(value, spatial_shapes, level_start_index,
                                            reference_points, sampling_offsets, attn_logits)
    if grad_output.dtype != value.dtype or grad_output.numel() != n * lq * m * d:
        raise RuntimeError("grad_output must be [N, Lq, M*D] with value's dtype")
    _check_inputs((("grad_output", grad_output),))
    lib = _load()
    with torch.cuda.device(value.device):
        grad_value = torch.empty(value.shape, dtype=torch.float32, device=value.device)   # fp32 also for bf16
        grad_offsets = torch.empty_like(sampling_offsets)
        grad_logits = torch.empty_like(attn_logits)
        nbytes = int(lib.msda_fused_backward_workspace_bytes(n, s, m, d, nl, lq, npt))
        workspace = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=value.device)
        rc = getattr(lib, "msda_fused_backward_" + _SUFFIX[value.dtype])(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            reference_points.data_ptr(), sampling_offsets.data_ptr(), attn_logits.data_ptr(),
            grad_output.data_ptr(), n, s, m, d, nl, lq, npt, grad_value.data_ptr(),
            grad_offsets.data_ptr(), grad_logits.data_ptr(), workspace.data_ptr(), nbytes,
            torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        _raise(lib, "ms_deform_attn_fused_backward")
    return [grad_value, grad_offsets, grad_logits]
