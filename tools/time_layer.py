"""Diagnostic: one MSDeformAttn module forward+backward at the C2 shape, default vs fused prologue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instaorder_amd import MSDeformAttn, synthetic
torch.manual_seed(0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = synthetic.encoder_shapes(1024)
ss = torch.tensor(shapes, dtype=torch.int64).cuda()
lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
s = int(ss.prod(1).sum())
mod = MSDeformAttn(256, 3, 8, 4).cuda()
query = torch.randn(batch, s, 256, device="cuda", requires_grad=True)
src = torch.randn(batch, s, 256, device="cuda", requires_grad=True)
ref = synthetic.reference_points(shapes).cuda()[None, :, None, :].expand(batch, s, 3, 2).contiguous()
for fused in (False, True, False, True):
    mod.fused_prologue = fused
    for _ in range(3):
        mod(query, ref, src, ss, lsi).sum().backward()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 10
    e[0].record()
    for _ in range(n):
        out = mod(query, ref, src, ss, lsi)
    e[1].record()
    for _ in range(n):
        out = mod(query, ref, src, ss, lsi)
        out.sum().backward()
    e[2].record(); e[2].synchronize()
    f = e[0].elapsed_time(e[1]) / n
    fb = e[1].elapsed_time(e[2]) / n
    print("fused=%-5s module forward %.3f ms, forward+backward %.3f ms, peak memory %.2f GB" % (
        fused, f, fb, torch.cuda.max_memory_allocated() / 2**30))
