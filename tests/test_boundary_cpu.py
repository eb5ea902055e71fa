"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/msda_hip.h declares, and the Python-visible extension module behaves like the
reference's on a GPU-less host (importable; calls on CPU tensors raise RuntimeError --
reference ops/src/ms_deform_attn.h:43,65).  No compute calls are made here."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT

LIB = os.path.join(ROOT, "instaorder_amd", "libmsda_hip.so")
HEADER = os.path.join(ROOT, "include", "msda_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msda_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("msda_forward_f32", "msda_forward_f64", "msda_backward_f32", "msda_backward_f64",
              "msda_last_error", "msda_launch_count", "msda_abi_version"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "libmsda_hip.so missing: run python -m instaorder_amd.build_ext"
    lib = ctypes.CDLL(LIB)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    lib.msda_abi_version.restype = ctypes.c_int
    assert lib.msda_abi_version() == 5
    lib.msda_last_error.restype = ctypes.c_char_p
    assert lib.msda_last_error() == b""


def test_library_rejects_bad_dimensions_without_touching_the_gpu():
    lib = ctypes.CDLL(LIB)
    lib.msda_last_error.restype = ctypes.c_char_p
    args = [ctypes.c_void_p(0)] * 5 + [ctypes.c_int(x) for x in (1, 4, 0, 32, 1, 4, 4)] + \
        [ctypes.c_void_p(0), ctypes.c_void_p(0)]
    assert lib.msda_forward_f32(*args) == 1        # MSDA_ERR_INVALID_ARGUMENT (num_heads = 0)
    assert b"bad dimensions" in lib.msda_last_error()


def test_extension_module_importable_and_raises_on_cpu():
    import MultiScaleDeformableAttention as MSDA
    value = torch.zeros(1, 6, 2, 4)
    shapes = torch.tensor([[2, 3]])
    lsi = torch.tensor([0])
    loc = torch.zeros(1, 3, 2, 1, 2, 2)
    aw = torch.zeros(1, 3, 2, 1, 2)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 128)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, torch.zeros(1, 3, 8), 128)


def test_function_has_reference_signature():
    import inspect
    from instaorder_amd import MSDeformAttnFunction
    params = list(inspect.signature(MSDeformAttnFunction.forward).parameters)
    assert params == ["ctx", "value", "value_spatial_shapes", "value_level_start_index",
                      "sampling_locations", "attention_weights", "im2col_step"]
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDeformAttnFunction.apply(torch.zeros(1, 6, 2, 4), torch.tensor([[2, 3]]),
                                   torch.tensor([0]), torch.zeros(1, 3, 2, 1, 2, 2),
                                   torch.zeros(1, 3, 2, 1, 2), 128)


def test_product_path_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the product packages may name it."""
    for pkg in ("instaorder_amd", "MultiScaleDeformableAttention"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                    assert "msda_oracle" not in src, f
