"""instaorder_amd -- MI355X-native multi-scale deformable attention (MSDA), the one native
hot path of SNU-VGILab/InstaOrder (Mask2Former's MSDeformAttnPixelDecoder encoder).

  csrc/msda_hip.hip   hand-written HIP kernels (gfx950) + the C ABI (include/msda_hip.h)
  ext.py              host side of the extension boundary (checks, allocation, stream)
  functions.py        MSDeformAttnFunction (autograd wrapper), as ops/functions in the reference
"""
from .functions import MSDeformAttnFunction

__all__ = ["MSDeformAttnFunction"]
