"""instaorder_amd -- MI355X-native multi-scale deformable attention (MSDA), the one native
hot path of SNU-VGILab/InstaOrder (Mask2Former's MSDeformAttnPixelDecoder encoder).

  csrc/msda_hip.hip   host side + the C ABI (include/msda_hip.h); kernels in csrc/msda_*.inc (gfx950 HIP)
  ext.py              host side of the extension boundary (checks, allocation, stream)
  functions.py        MSDeformAttnFunction (autograd wrapper) + the CPU-tensor PyTorch path
  modules.py          MSDeformAttn (the nn.Module that calls the op)
  encoder.py          the pixel decoder's MSDA encoder stack, detectron2-free
  pixel_decoder.py    MSDeformAttnPixelDecoder around it (input projections, FPN, mask features)
  synthetic.py        synthetic inputs of the encoder's shape (bench, tests)
"""
from .functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch
from .modules import MSDeformAttn

__all__ = ["MSDeformAttnFunction", "MSDeformAttn", "ms_deform_attn_core_pytorch"]
