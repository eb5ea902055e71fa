"""CPU: the MSDeformAttn module mirror against the reference module's golden output, the
encoder stack's structure, and the world_size-2 (gloo) data-parallel path."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import cases
from conftest import ROOT, load_golden, rel_err


def _module_and_inputs(dtype):
    from instaorder_amd import MSDeformAttn
    mc = cases.make_module_case()
    mod = MSDeformAttn(mc["d_model"], mc["n_levels"], mc["n_heads"], mc["n_points"]).to(dtype)
    mod.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in mc["state_dict"].items()})
    t = lambda k: torch.from_numpy(mc[k]).to(dtype)
    return mod, mc, t


def test_state_dict_keys_match_reference():
    from instaorder_amd import MSDeformAttn
    keys = sorted(MSDeformAttn(32, 2, 4, 2).state_dict().keys())
    assert keys == sorted(cases.make_module_case()["state_dict"].keys())


def test_module_cpu_matches_reference_golden():
    mod, mc, t = _module_and_inputs(torch.float64)
    g = load_golden("module_msdeformattn")
    query, src = t("query").requires_grad_(True), t("src").requires_grad_(True)
    out = mod(query, t("ref"), src, torch.from_numpy(mc["shapes"]), torch.from_numpy(mc["lsi"]),
              torch.from_numpy(mc["padding_mask"]))
    up = torch.from_numpy(np.random.RandomState(202).standard_normal(tuple(out.shape)).astype(np.float32)).double()
    out.backward(up)
    assert rel_err(out.detach().numpy(), g["out"]) < 1e-12
    assert rel_err(query.grad.numpy(), g["grad_query"]) < 1e-11
    assert rel_err(src.grad.numpy(), g["grad_src"]) < 1e-11


def test_encoder_stack_structure_and_grad_flow():
    from instaorder_amd.encoder import MSDeformAttnTransformerEncoderOnly, PositionEmbeddingSine
    torch.manual_seed(0)
    enc = MSDeformAttnTransformerEncoderOnly(d_model=64, nhead=2, num_encoder_layers=2,
                                             dim_feedforward=128, dropout=0.0, num_feature_levels=3,
                                             use_adapters=True)
    keys = enc.state_dict().keys()
    for k in ("level_embed", "encoder.layers.0.self_attn.sampling_offsets.weight",
              "encoder.layers.1.self_attn.value_proj.bias", "encoder.layers.0.norm2.weight",
              "encoder.layers.1.adapter.2.weight", "encoder.layers.0.linear1.weight"):
        assert k in keys, k
    pe = PositionEmbeddingSine(32, normalize=True)
    srcs = [torch.randn(2, 64, h, w) for h, w in [(2, 3), (4, 6), (8, 12)]]      # low-res first
    memory, shapes, lsi = enc(srcs, [pe(x) for x in srcs])
    assert tuple(memory.shape) == (2, 126, 64)
    assert shapes.tolist() == [[2, 3], [4, 6], [8, 12]] and lsi.tolist() == [0, 6, 30]
    memory.sum().backward()
    assert all(p.grad is not None for p in enc.parameters())
    ref = enc.encoder.get_reference_points(shapes, torch.ones(2, 3, 2), "cpu")
    assert tuple(ref.shape) == (2, 126, 3, 2)
    assert torch.allclose(ref[0, 0, 0], torch.tensor([0.5 / 3, 0.5 / 2]))       # first pixel centre (x, y)


@pytest.mark.parametrize("key,kw,shape,masked", cases.POSENC_CASES)
def test_position_embedding_sine_matches_reference_golden(key, kw, shape, masked):
    """instaorder_amd.encoder.PositionEmbeddingSine against outputs of the reference's own class
    (transformer_decoder/position_encoding.py:12-56, loaded by file path in make_golden.py)."""
    from instaorder_amd.encoder import PositionEmbeddingSine
    want = load_golden("position_encoding")[key]
    got = PositionEmbeddingSine(**kw)(torch.zeros(shape), cases.posenc_mask(shape) if masked else None)
    assert got.shape == want.shape
    assert float((got - torch.from_numpy(want)).abs().max()) < 1e-6


def _encoder_case(key, dtype, device):
    from instaorder_amd.encoder import MSDeformAttnTransformerEncoderOnly
    ec, g = cases.make_encoder_case(key), load_golden("encoder_" + key)
    enc = MSDeformAttnTransformerEncoderOnly(**ec["kwargs"]).to(dtype).to(device)
    enc.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in ec["state_dict"].items()})   # strict: same keys
    enc.eval()
    srcs = [torch.from_numpy(x).to(dtype).to(device).requires_grad_(True) for x in ec["srcs"]]
    pos = [torch.from_numpy(x).to(dtype).to(device) for x in ec["pos"]]
    memory, shapes, lsi = enc(srcs, pos)
    memory.backward(torch.from_numpy(ec["grad_memory"]).to(dtype).to(device))
    return g, memory, shapes, lsi, srcs


@pytest.mark.parametrize("key", sorted(cases.ENCODER_CASES))
def test_encoder_stack_matches_reference_golden(key):
    """Row f1: instaorder_amd.encoder.MSDeformAttnTransformerEncoderOnly (fp64, CPU path) against the output
    and input gradients of the REFERENCE's class (pixel_decoder/msdeformattn.py:23-254, run by
    tests/golden/make_golden.py) with the same seeded parameters -- loaded under the reference's state-dict
    keys, adapters included."""
    g, memory, shapes, lsi, srcs = _encoder_case(key, torch.float64, "cpu")
    tol = 1e-12 if g["memory"].dtype == np.float64 else 1e-6          # 'heads8' is stored in fp32
    assert (shapes.numpy() == g["shapes"]).all() and (lsi.numpy() == g["lsi"]).all()
    assert rel_err(memory.detach().numpy(), g["memory"]) < tol
    for i, x in enumerate(srcs):
        assert rel_err(x.grad.numpy(), g["grad_src%d" % i]) < tol


def test_shard_batch_covers_the_batch():
    from instaorder_amd.distributed import shard_batch
    for world in (1, 2, 3, 8):
        spans = [shard_batch(16, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 16
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_data_parallel_two_ranks_gloo(tmp_path):
    """world_size 2 over gloo: the harness runs, both ranks agree on the result, and sharding
    the batch over two ranks leaves the loss of the first step unchanged (same global batch)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    common = ["--cpu", "--image", "64", "--global-batch", "4", "--steps", "1", "--warmup", "0"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "train_bench.py")] + common,
                         capture_output=True, text=True, env=env, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
                          os.path.join(ROOT, "train_bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, env=env, timeout=300)
    assert two.returncode == 0, two.stderr[-2000:]
    import json
    r1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    r2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert r1["n_gpus"] == 1 and r2["n_gpus"] == 2
    assert r2["unit"] == "imgs/s" and r2["value"] > 0


def test_train_bench_self_launches_or_refuses():
    """`python train_bench.py --gpus 2` without a launcher starts its own two ranks (gloo here); with a
    WORLD_SIZE that contradicts --gpus it refuses instead of reporting one rank as two."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMP_NUM_THREADS="2")
    common = ["--cpu", "--image", "64", "--global-batch", "4", "--steps", "1", "--warmup", "0"]
    run = subprocess.run([sys.executable, os.path.join(ROOT, "train_bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    r = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["value"] > 0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "train_bench.py"), "--gpus", "4"] + common,
                         capture_output=True, text=True, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_bench_self_launches_two_ranks_gloo():
    """`python bench.py --gpus 2` with no launcher in the environment starts its own two ranks
    (fresh child processes under torch.distributed.run).  On this GPU-less host the ranks run the
    control path only (MSDA_BENCH_REHEARSAL=1: gloo rendezvous, barriers, max over ranks, one JSON
    line from rank 0, no kernels) -- what the driver's `--gpus N` command exercises around the
    timed region."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MSDA_BENCH_REHEARSAL="1", OMP_NUM_THREADS="2")
    if torch.cuda.is_available():
        env["MSDA_BENCH_BACKEND"] = "gloo"      # two ranks sharing one GPU cannot use RCCL
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--workload", "tiny", "--no-cpu-baseline", "--no-extras"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["unit"] == "Gsamples/s"
    # without a GPU and without the rehearsal switch the ranks refuse to run: no CPU fallback
    if not torch.cuda.is_available():
        env.pop("MSDA_BENCH_REHEARSAL")
        bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                             capture_output=True, text=True, env=env, timeout=120)
        assert bad.returncode != 0 and "no CPU fallback" in bad.stderr


def test_config_c1_cpu_forward_512():
    """BASELINE.json configs[0]: CPU forward at the 512^2 shape (levels 16^2/32^2/64^2, S=5376)
    goes through the PyTorch path of the module -- the reference's CPU behaviour -- and agrees
    with the C oracle on the op's inputs."""
    from instaorder_amd import MSDeformAttn, synthetic, ms_deform_attn_core_pytorch
    from oracle import msda_oracle
    torch.manual_seed(0)
    shapes = synthetic.encoder_shapes(512)
    inp = synthetic.make_inputs(1, shapes, dist="model", seed=3)
    out = ms_deform_attn_core_pytorch(inp["value"], inp["spatial_shapes"], inp["sampling_locations"],
                                      inp["attention_weights"])
    ref = msda_oracle.forward(inp["value"].numpy(), inp["spatial_shapes"].numpy(),
                              inp["level_start_index"].numpy(), inp["sampling_locations"].numpy(),
                              inp["attention_weights"].numpy())
    assert rel_err(out.numpy(), ref) < 1e-5
    mod = MSDeformAttn(256, 3, 8, 4)
    s = sum(h * w for h, w in shapes)
    x = torch.randn(1, s, 256)
    ref_pts = synthetic.reference_points(shapes)[None, :, None, :].expand(1, s, 3, 2)
    y = mod(x, ref_pts, x, inp["spatial_shapes"], inp["level_start_index"])
    assert tuple(y.shape) == (1, s, 256) and torch.isfinite(y).all()


def test_pixel_decoder_mirror_shapes_and_state_dict_keys():
    """Row f4: the detectron2-free MSDeformAttnPixelDecoder (reference msdeformattn.py:257-487):
    output shapes of forward_features, the reference's sub-module names, gradients reach the
    input projections through the encoder."""
    import torch
    from instaorder_amd.pixel_decoder import MSDeformAttnPixelDecoder
    torch.manual_seed(0)
    shape = {"res2": (96, 4), "res3": (192, 8), "res4": (384, 16), "res5": (768, 32)}
    pd = MSDeformAttnPixelDecoder(shape, transformer_enc_layers=2, use_adapters=True)
    keys = set(pd.state_dict().keys())
    for k in ("input_proj.0.0.weight", "input_proj.2.1.bias", "mask_features.weight", "mask_features.bias",
              "adapter_1.weight", "adapter_1.norm.weight", "layer_1.weight", "layer_1.norm.bias",
              "transformer.level_embed", "transformer.encoder.layers.0.self_attn.sampling_offsets.weight",
              "transformer.encoder.layers.1.self_attn.value_proj.bias"):
        assert k in keys, k
    assert "adapter_1.bias" not in keys                    # norm == "GN": convs without bias
    feats = {k: torch.randn(2, c, 96 // s, 128 // s) for k, (c, s) in shape.items()}
    mask_features, enc_low, multi = pd.forward_features(feats)
    assert mask_features.shape == (2, 256, 24, 32)         # stride 4
    assert enc_low.shape == (2, 256, 3, 4)                 # stride 32
    assert [tuple(m.shape[-2:]) for m in multi] == [(3, 4), (6, 8), (12, 16)]
    (mask_features.square().mean() + multi[2].mean()).backward()
    assert pd.input_proj[0][0].weight.grad is not None and pd.input_proj[0][0].weight.grad.abs().sum() > 0
    assert pd.adapter_1.weight.grad.abs().sum() > 0
