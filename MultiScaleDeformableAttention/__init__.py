"""Drop-in for the reference's compiled module of the same name.

The reference imports it at module scope
(mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:21-29) and calls
``MSDA.ms_deform_attn_forward`` / ``MSDA.ms_deform_attn_backward`` (:36-37, :45-47), which its
pybind layer exports (ops/src/vision.cpp:18-21).  With this repository's root on
``sys.path`` the reference's ops/functions, ops/modules and msdeformattn.py run unchanged
on MI355X.  Importable on a GPU-less host (calls then raise), as the reference's is.
"""
from instaorder_amd.ext import ms_deform_attn_backward, ms_deform_attn_forward

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
