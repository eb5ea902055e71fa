"""Generates tests/golden/*.npz by running the REFERENCE's own Python oracle.

Run in the build container only (needs /root/reference; the GPU box has none):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What runs: ``ms_deform_attn_core_pytorch`` (reference
mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:52-72)
in float64 on the seeded inputs of cases.py, with float64 autograd for the three
gradients, and -- for the module fixture -- the reference ``MSDeformAttn``
module (ops/modules/ms_deform_attn.py:34-125) on CPU, which reaches the same
function through its bare ``except`` (:119-121).  The reference imports a
compiled extension at module scope (ms_deform_attn_func.py:21-29); an empty
stand-in module of that name is registered first so the import succeeds -- the
extension itself is never called on this path.

Only the OUTPUTS are stored; inputs are re-created from cases.py's seeds.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

REF_OPS_PARENT = "/root/reference/mask2former/modeling/pixel_decoder"


def import_reference():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("MultiScaleDeformableAttention",
                           types.ModuleType("MultiScaleDeformableAttention"))
    sys.path.insert(0, REF_OPS_PARENT)
    from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    from ops.modules.ms_deform_attn import MSDeformAttn
    return ms_deform_attn_core_pytorch, MSDeformAttn


REF_POSENC = "/root/reference/mask2former/modeling/transformer_decoder/position_encoding.py"


def position_encoding_golden():
    """PositionEmbeddingSine (reference transformer_decoder/position_encoding.py:12-56; the file
    imports nothing but torch), loaded by file path, on the encoder's configuration
    (num_pos_feats = 128, normalize = True, pixel_decoder/msdeformattn.py:311-312) with and without
    a padding mask, and on the un-normalised default."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_position_encoding", REF_POSENC)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for key, kw, shape, masked in cases.POSENC_CASES:
        pe = mod.PositionEmbeddingSine(**kw)
        x = torch.zeros(shape)
        out[key] = pe(x, cases.posenc_mask(shape) if masked else None).numpy()
    path = os.path.join(HERE, "position_encoding.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def import_reference_encoder():
    """The reference's encoder classes (pixel_decoder/msdeformattn.py:23-254) without detectron2.

    msdeformattn.py imports detectron2 / fvcore NAMES at module scope (configurable, Conv2d, ShapeSpec,
    get_norm, SEM_SEG_HEADS_REGISTRY, weight_init) that only the pixel-decoder class below the encoder
    classes uses, and sits in the `mask2former` package whose __init__ pulls in all of detectron2.  Empty
    placeholder modules are registered for those names and for the enclosing packages (their __path__ set to
    the reference's directories so that the real position_encoding.py, transformer.py and ops/ files load);
    nothing the encoder classes execute comes from a placeholder.  Returns the module."""
    import importlib
    sys.dont_write_bytecode = True
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    root = "/root/reference/mask2former"

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    pkg("mask2former", root)
    pkg("mask2former.modeling", root + "/modeling")
    pkg("mask2former.modeling.pixel_decoder", root + "/modeling/pixel_decoder")
    pkg("mask2former.modeling.transformer_decoder", root + "/modeling/transformer_decoder")

    class _Registry:
        def register(self):
            return lambda cls: cls

    def placeholder(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    placeholder("fvcore")
    placeholder("fvcore.nn", weight_init=placeholder("fvcore.nn.weight_init"))
    placeholder("detectron2")
    placeholder("detectron2.config", configurable=lambda f=None, **kw: f if f is not None else (lambda g: g))
    placeholder("detectron2.layers", Conv2d=None, ShapeSpec=None, get_norm=None)
    placeholder("detectron2.modeling", SEM_SEG_HEADS_REGISTRY=_Registry())
    return importlib.import_module("mask2former.modeling.pixel_decoder.msdeformattn")


def encoder_golden():
    """MSDeformAttnTransformerEncoderOnly of the reference (pixel_decoder/msdeformattn.py:23-123, with its
    encoder layers :126-206 and get_reference_points :216-229), fp64 on the CPU -- the op inside takes the
    reference's ms_deform_attn_core_pytorch path -- on the seeded inputs and parameters of
    cases.make_encoder_case(): output memory and the gradients of the inputs."""
    ref = import_reference_encoder()
    for key in cases.ENCODER_CASES:
        ec = cases.make_encoder_case(key)
        enc = ref.MSDeformAttnTransformerEncoderOnly(**ec["kwargs"]).double()
        missing = set(enc.state_dict().keys()) ^ set(ec["state_dict"].keys())
        assert not missing, missing
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in ec["state_dict"].items()})
        enc.eval()
        srcs = [torch.from_numpy(x).requires_grad_(True) for x in ec["srcs"]]
        pos = [torch.from_numpy(x) for x in ec["pos"]]
        memory, shapes, lsi = enc(srcs, pos)
        memory.backward(torch.from_numpy(ec["grad_memory"]))
        out = {"memory": memory.detach().numpy(), "shapes": shapes.numpy(), "lsi": lsi.numpy()}
        for i, x in enumerate(srcs):
            out["grad_src%d" % i] = x.grad.numpy()
        if key in ("heads8", "six_layers"):        # compared in fp32 on the GPU: stored in fp32 to stay small
            out = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in out.items()}
        path = os.path.join(HERE, "encoder_%s.npz" % key)
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes")


def main():
    position_encoding_golden()
    if "--encoder-only" in sys.argv:
        return encoder_golden()
    if "--posenc-only" in sys.argv:
        return
    core, MSDeformAttn = import_reference()
    torch.set_num_threads(4)
    for name in cases.CASES:
        c = cases.make_case(name)
        value = torch.from_numpy(c["value"]).requires_grad_(True)
        loc = torch.from_numpy(c["loc"]).requires_grad_(True)
        aw = torch.from_numpy(c["aw"]).requires_grad_(True)
        shapes = torch.from_numpy(c["shapes"])
        out = core(value, shapes, loc, aw)
        out.backward(torch.from_numpy(c["grad_out"]))
        path = os.path.join(HERE, name + ".npz")
        np.savez(path, out=out.detach().numpy(), grad_value=value.grad.numpy(),
                 grad_loc=loc.grad.numpy(), grad_aw=aw.grad.numpy())
        print("wrote", path, os.path.getsize(path), "bytes")

    mc = cases.make_module_case()
    mod = MSDeformAttn(d_model=mc["d_model"], n_levels=mc["n_levels"],
                       n_heads=mc["n_heads"], n_points=mc["n_points"]).double()
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in mc["state_dict"].items()})
    query = torch.from_numpy(mc["query"]).requires_grad_(True)
    src = torch.from_numpy(mc["src"]).requires_grad_(True)
    out = mod(query, torch.from_numpy(mc["ref"]), src, torch.from_numpy(mc["shapes"]),
              torch.from_numpy(mc["lsi"]), torch.from_numpy(mc["padding_mask"]))
    g = torch.from_numpy(np.random.RandomState(202).standard_normal(tuple(out.shape)).astype(np.float32)).double()
    out.backward(g)
    path = os.path.join(HERE, "module_msdeformattn.npz")
    np.savez(path, out=out.detach().numpy(), grad_query=query.grad.numpy(), grad_src=src.grad.numpy())
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
