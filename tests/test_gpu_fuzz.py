"""GPU fuzz of the D == 32 fast paths against the C oracle: random level pyramids, heads, points,
query counts and batch sizes (run on the MI355X box with -m gpu).  Sizes keep the oracle fast;
the last cases are shaped so that the backward's tile lists are long enough to be split into
parts (small batch, many queries per tile)."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

FWD_TOL_F32 = 1e-5
GRAD_TOL_F32 = 1e-4


def _cases():
    rng = np.random.RandomState(1234)
    cases = []
    for i in range(24):
        nl = int(rng.randint(1, 5))
        shapes = [(int(rng.randint(1, 41)), int(rng.randint(1, 41))) for _ in range(nl)]
        heads = int(rng.choice([1, 2, 3, 5, 8]))
        points = int(rng.randint(1, 5))
        batch = int(rng.randint(1, 4))
        s = sum(h * w for h, w in shapes)
        num_query = None if i % 3 == 0 else int(rng.randint(1, 400))
        spread = float(rng.choice([1.0, 1.3, 3.0]))
        cases.append((shapes, heads, points, batch, num_query, spread, 100 + i))
    # long lists on few tiles: the split path of the tile kernel
    cases.append(([(16, 16)], 2, 4, 1, 4000, 1.1, 7))
    cases.append(([(20, 30), (9, 12)], 3, 4, 2, 2500, 1.0, 8))
    cases.append(([(5, 7), (33, 18), (64, 64)], 8, 3, 1, 3000, 1.05, 9))
    # many levels / points: up to 8 levels stay on the binned backward; 9 levels, or more than 4
    # points, take the generic backward; more than 47 samples per pair the plain D == 32 forward
    pyr = [(24, 31), (12, 16), (6, 8), (3, 4), (2, 2), (1, 3), (5, 1), (1, 1), (7, 2)]
    cases.append((pyr[:8], 2, 4, 2, 300, 1.2, 10))
    cases.append((pyr[:6], 3, 1, 1, None, 1.0, 11))
    cases.append((pyr[:9], 2, 2, 2, 150, 1.3, 12))
    cases.append((pyr[:4], 2, 12, 1, 200, 1.1, 13))
    cases.append((pyr[:2], 5, 8, 2, None, 1.0, 14))
    return cases


@pytest.mark.parametrize("shapes,heads,points,batch,num_query,spread,seed", _cases())
def test_fuzz_d32_vs_oracle(oracle, shapes, heads, points, batch, num_query, spread, seed):
    import MultiScaleDeformableAttention as MSDA
    from instaorder_amd import ext, synthetic
    kw = {} if num_query is None else {"num_query": num_query}
    inp = synthetic.make_inputs(batch, shapes, heads=heads, channels=32, points=points, dist="uniform",
                                seed=seed, **kw)
    loc = inp["sampling_locations"]
    loc = (loc - 0.5) * spread + 0.5                              # some samples outside the levels
    # stay off the pixel lattice, where fp32 and fp64 may floor differently (see test_gpu_parity)
    for l, (h, w) in enumerate(shapes):
        for c, size in ((0, w), (1, h)):
            pix = loc[..., l, :, c].double() * size - 0.5
            near = (pix - pix.round()).abs() < 1e-3
            loc[..., l, :, c] += near.float() * (3e-3 / size)
    inp["sampling_locations"] = loc.contiguous()
    dev = {k: v.cuda() for k, v in inp.items()}
    before = ext.launch_count()
    out = MSDA.ms_deform_attn_forward(dev["value"], dev["spatial_shapes"], dev["level_start_index"],
                                      dev["sampling_locations"], dev["attention_weights"], 128)
    gv, gl, ga = MSDA.ms_deform_attn_backward(dev["value"], dev["spatial_shapes"],
                                              dev["level_start_index"], dev["sampling_locations"],
                                              dev["attention_weights"], dev["grad_output"], 128)
    torch.cuda.synchronize()
    assert ext.launch_count() >= before + 2
    n = lambda k: inp[k].double().numpy() if inp[k].is_floating_point() else inp[k].numpy()
    args = (n("value"), n("spatial_shapes"), n("level_start_index"), n("sampling_locations"),
            n("attention_weights"))
    o_out = oracle.forward(*args)
    o_gv, o_gl, o_ga = oracle.backward(*args, n("grad_output"))
    assert rel_err(out.cpu().numpy(), o_out) < FWD_TOL_F32
    assert rel_err(gv.cpu().numpy(), o_gv) < GRAD_TOL_F32
    assert rel_err(gl.cpu().numpy(), o_gl) < GRAD_TOL_F32
    assert rel_err(ga.cpu().numpy(), o_ga) < GRAD_TOL_F32
