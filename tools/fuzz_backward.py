"""Diagnostic (GPU box): random D == 32 problems through the C ABI's backward entry points with random workspace
sizes between the minimum and the whole batch's worst case, checked against the C oracle in fp64.
python tools/fuzz_backward.py [cases] [seed].  tools/ is not part of the product; this is the only place outside
tests/ that uses the oracle, as a checker."""
import ctypes, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from instaorder_amd import ext, synthetic
from oracle import msda_oracle

msda_oracle.build()
lib = ext._load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def off_lattice(inp):
    loc = inp["sampling_locations"]
    for l, (h, w) in enumerate(inp["spatial_shapes"].tolist()):
        for c, size in ((0, w), (1, h)):
            pix = loc[..., l, :, c].double() * size - 0.5
            near = (pix - pix.round()).abs() < 1e-3
            loc[..., l, :, c] += near.float() * (3e-3 / size)
    return inp


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


worst = 0.0
for case in range(cases):
    nl = rng.randint(1, 4)
    shapes = [(rng.randint(1, 70), rng.randint(1, 70)) for _ in range(nl)]
    heads, points, batch = rng.choice([1, 2, 3, 8]), rng.randint(1, 4), rng.randint(1, 5)
    s = sum(h * w for h, w in shapes)
    lq = rng.choice([1, 7, 100, 500, s])
    spread = rng.choice([1.0, 1.3, 3.0])
    bf16 = rng.random() < 0.4
    inp = synthetic.make_inputs(batch, shapes, heads=heads, channels=32, points=points, dist="uniform",
                                seed=rng.randint(0, 1 << 30), num_query=lq)
    inp["sampling_locations"] = (inp["sampling_locations"] - 0.5) * spread + 0.5
    if rng.random() < 0.2:      # pile everything onto a few pixels
        inp["sampling_locations"] = inp["sampling_locations"] * 0.02 + rng.random() * 0.9
    inp = off_lattice(inp)
    if bf16:
        inp["value"] = inp["value"].bfloat16().float()
        inp["grad_output"] = inp["grad_output"].bfloat16().float()
    nn = lambda k: inp[k].double().numpy() if inp[k].is_floating_point() else inp[k].numpy()
    o = msda_oracle.backward(nn("value"), nn("spatial_shapes"), nn("level_start_index"), nn("sampling_locations"),
                             nn("attention_weights"), nn("grad_output"))
    dev = {k: t.cuda() for k, t in inp.items()}
    v, go = dev["value"], dev["grad_output"]
    gv16 = bf16 and rng.random() < 0.6
    if bf16:
        v, go = v.bfloat16(), go.bfloat16()
    loc, aw = dev["sampling_locations"], dev["attention_weights"]
    dims = (batch, s, heads, 32, nl, lq, points)
    full = int((lib.msda_backward_bf16_gv16_workspace_bytes if gv16 else lib.msda_backward_workspace_bytes)(*dims))
    least = int(lib.msda_backward_min_workspace_bytes(*dims))
    nbytes = rng.choice([full, least, rng.randint(least, full), rng.randint(least, least + (full - least) // 4)])
    guard = 1024
    buf = torch.full((guard + nbytes + guard + 16,), 0x5A, dtype=torch.uint8, device="cuda")
    base = buf.data_ptr() + guard
    base += (-base) % 16
    off0 = base - buf.data_ptr()
    gv = torch.full(v.shape, 7.0, device="cuda", dtype=torch.bfloat16 if gv16 else torch.float32)
    gl, ga = torch.full_like(loc, 7.0), torch.full_like(aw, 7.0)
    fn = lib.msda_backward_bf16_gv16_ws if gv16 else (lib.msda_backward_bf16_ws if bf16 else lib.msda_backward_f32_ws)
    rc = fn(v.data_ptr(), dev["spatial_shapes"].data_ptr(), dev["level_start_index"].data_ptr(), loc.data_ptr(),
            aw.data_ptr(), go.data_ptr(), *dims, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), base, nbytes,
            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.msda_last_error()
    torch.cuda.synchronize()
    ok_guard = bool((buf[:off0] == 0x5A).all()) and bool((buf[off0 + nbytes:] == 0x5A).all())
    e = (rel(gv.float().cpu().numpy().astype(np.float64), o[0]), rel(gl.cpu().numpy().astype(np.float64), o[1]),
         rel(ga.cpu().numpy().astype(np.float64), o[2]))
    tol = (4e-3 if gv16 else 1e-4, 1e-4, 1e-4)
    bad = not ok_guard or any(x > t for x, t in zip(e, tol))
    worst = max(worst, e[1], e[2], e[0] if not gv16 else 0.0)
    if bad or case % 20 == 0:
        print("%s case %d shapes %s M %d P %d N %d Lq %d spread %.1f %s ws %d of [%d, %d] guard %s err %.2e %.2e %.2e"
              % ("BAD" if bad else "ok ", case, shapes, heads, points, batch, lq, spread,
                 "gv16" if gv16 else ("bf16" if bf16 else "f32"), nbytes, least, full, ok_guard, *e), flush=True)
    if bad:
        sys.exit(1)
print("all %d cases ok, worst fp32-grade error %.2e" % (cases, worst))
