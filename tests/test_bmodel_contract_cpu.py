"""B-model boundary: the checkpoint contract of `MDQE` -- `load_state_dict(strict=True)` with the REFERENCE's key
set (parameter names recorded from the reference's own MDQE module in the fixture video_small, plus the aliased decoder
modules transformer_dec.py:37-46, the fixed buffers, criterion.* and BatchNorm counters a released .pth carries) -- and the
top-level `MultiScaleDeformableAttention` shim the reference imports at ops/functions/ms_deform_attn_func.py:19."""
import dataclasses
import re

import pytest
import torch

from _golden import Fixture
from mdqe_cvpr2023_amd.config import MDQEConfig
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import ALIASES, full_manifest, random_state

SMALL = dict(enc_layers=2, dec_layers=2, n_frames=3, num_classes=5, num_queries=16, query_embed_dim=16)


def _reference_keys():
    fx = Fixture("video_small")
    return {str(n) for n in fx.z["manifest_names"]}


def test_manifest_names_are_the_reference_modules_names():
    """Everything outside the (third-party) backbone: the product's parameter names + the aliases == the names the reference's
    MDQE module registered when the fixture was made."""
    ref = {k for k in _reference_keys() if ".backbone.0.backbone." not in k}
    cfg = MDQEConfig(backbone="custom", backbone_channels=(16, 24, 32), **SMALL)
    mine = set(full_manifest(cfg))
    alias = {a + k[len(b):] for k in mine for a, b in ALIASES.items() if k.startswith(b)}
    assert mine | alias == ref, (sorted(ref - mine - alias)[:5], sorted((mine | alias) - ref)[:5])


def _released_checkpoint(cfg, seed):
    """A state dict shaped like a released .pth: every parameter, the aliased copies, the fixed buffers, training-only keys."""
    sd = random_state(cfg, seed=seed)
    out = dict(sd)
    for k, v in sd.items():
        for a, b in ALIASES.items():
            if k.startswith(b):
                out[a + k[len(b):]] = v                                       # transformer_dec.py:37-46: the same tensors twice
    nh, L, P = cfg.nheads, cfg.n_levels, cfg.dec_points
    for i in range(cfg.enc_layers):
        out[f"detr.transformer_enc.encoder.layers.{i}.self_attn.lvl_spatial_scales"] = torch.arange(1, L + 1).float()
    for i in range(cfg.dec_layers):
        q = f"detr.transformer_dec.decoder.layers.{i}"
        out[q + ".cross_attn.sampling_offsets"] = torch.zeros(1, 1, nh, L, P, 2)      # buffer, ms_deform_attn.py:81-87
        out[q + ".cross_attn.lvl_spatial_scales"] = torch.arange(1, L + 1).float()
        out[q + ".temp_attn_inst.sampling_offsets"] = torch.zeros(1, 1, nh, cfg.n_frames, P, 2)
        out[q + ".temp_attn_inst.lvl_spatial_scales"] = torch.full((cfg.n_frames,), 2.0)
    out["detr.transformer_dec.query_relpos_grid"] = torch.zeros(cfg.n_query, cfg.n_query, 2)
    out["criterion.empty_weight"] = torch.ones(cfg.num_classes + 1)
    for k in list(sd):
        if k.endswith("running_var"):
            out[k[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(0)
    return sd, out


def test_load_state_dict_strict_with_a_released_checkpoints_key_set():
    cfg = MDQEConfig(**SMALL)                                                  # R50 backbone
    sd, ckpt = _released_checkpoint(cfg, seed=5)
    model = MDQE(cfg, seed=1)                                                  # different weights
    assert any(not torch.equal(v, sd[k]) for k, v in model.state_dict().items())
    res = model.load_state_dict(ckpt, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    got = model.state_dict()
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v.float()), k
    via_ctor = MDQE(cfg, state_dict=ckpt).state_dict()                        # the constructor path takes the same dict
    assert all(torch.equal(via_ctor[k], got[k]) for k in got)
    assert model.eval() is model and model.training is False
    with pytest.raises(RuntimeError):
        model.train()


def test_swin_checkpoint_key_set_loads_strictly_into_the_model_and_into_the_backbone_builder():
    """A Swin checkpoint carries the blocks' fixed buffers (`relative_coords_table`, `relative_position_index`: persistent buffers of
    WindowAttention, swin_transformer_v2.py:120,133): accepted and dropped by `MDQE.load_state_dict(strict=True)` and by the
    `build_swinv2_backbone` module; every parameter name of the backbone module equals the reference's (fixture swin_small's manifest)."""
    from _golden import Fixture
    from mdqe_cvpr2023_amd import SwinTransformerV2
    swin = dict(backbone="SwinV2", swin_embed_dim=32, swin_depths=(2, 2, 2, 2), swin_heads=(2, 4, 8, 16), swin_window=4, backbone_channels=(64, 128, 256))
    cfg = MDQEConfig(**swin, **SMALL)
    sd, ckpt = _released_checkpoint(cfg, seed=4)
    bp = "detr.backbone.0.backbone."
    for i, depth in enumerate(cfg.swin_depths):
        ws = 4 // 2 if i == 3 else 4
        for j in range(depth):
            ckpt[f"{bp}layers.{i}.blocks.{j}.attn.relative_coords_table"] = torch.zeros(1, 2 * ws - 1, 2 * ws - 1, 2)
            ckpt[f"{bp}layers.{i}.blocks.{j}.attn.relative_position_index"] = torch.zeros(ws * ws, ws * ws, dtype=torch.long)
    res = MDQE(cfg, seed=1).load_state_dict(dict(ckpt), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert MDQE(cfg, state_dict=dict(ckpt)) is not None
    bb = SwinTransformerV2(embed_dim=32, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), window_size=4, device="cpu")
    fx = Fixture("swin_small")                                                 # parameter names of the REFERENCE's module
    assert set(bb.state_dict()) == {str(n).replace("bb.", "", 1) for n in fx.z["manifest_names"]}
    only = {k[len(bp):]: v for k, v in ckpt.items() if k.startswith(bp)}
    res = bb.load_state_dict(only, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(bb.state_dict()["layers.2.blocks.1.attn.qkv.weight"], sd[bp + "layers.2.blocks.1.attn.qkv.weight"])
    assert bb.output_shape()["stage4"].channels == 128 and bb.output_shape()["stage4"].stride == 16 and bb.size_divisibility == 32
    assert bb.train(False) is bb
    with pytest.raises(RuntimeError):
        bb.train()


def test_only_aliased_decoder_copies_present():
    """A checkpoint that carries the shared decoder modules ONLY under the alias (decoder.bbox_embed ...) still loads."""
    cfg = MDQEConfig(**SMALL)
    sd, ckpt = _released_checkpoint(cfg, seed=2)
    for b in ALIASES.values():
        for k in [k for k in ckpt if k.startswith(b)]:
            del ckpt[k]
    m = MDQE(cfg, state_dict=ckpt)
    assert torch.equal(m.state_dict()["detr.transformer_dec.bbox_embed.layers.0.weight"], sd["detr.transformer_dec.bbox_embed.layers.0.weight"])


def test_missing_or_foreign_keys_are_loud():
    cfg = MDQEConfig(**SMALL)
    sd, ckpt = _released_checkpoint(cfg, seed=2)
    bad = dict(ckpt)
    del bad["detr.transformer_enc.encoder.layers.0.linear1.weight"]
    with pytest.raises(RuntimeError, match="Missing key"):
        MDQE(cfg, seed=0).load_state_dict(bad, strict=True)
    with pytest.raises(KeyError):
        MDQE(cfg, state_dict=bad)                                              # no silent zero-filled layer
    bad = dict(ckpt)
    bad["detr.transformer_enc.encoder.layers.0.linear9.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        MDQE(cfg, seed=0).load_state_dict(bad, strict=True)
    wrong = dict(ckpt)
    wrong["detr.transformer_enc.level_embed"] = torch.zeros(3, 7)
    with pytest.raises(RuntimeError, match="size mismatch"):
        MDQE(cfg, seed=0).load_state_dict(wrong, strict=True)


def test_merge_on_cpu_is_read_from_the_config():
    from types import SimpleNamespace as NS
    from test_config_cpu import _cfg
    from mdqe_cvpr2023_amd.config import PRESETS, from_d2_cfg
    assert from_d2_cfg(_cfg()).merge_on_cpu is True                           # (the helper's tree sets MERGE_ON_CPU=True)
    assert PRESETS["R50_ovis_360"].merge_on_cpu is False and PRESETS["R50_ovis_720"].merge_on_cpu is True
    assert PRESETS["swinl_ovis"].merge_on_cpu is True


def test_top_level_msda_shim_exports_the_extensions_two_functions():
    import MultiScaleDeformableAttention as MSDA                              # what ms_deform_attn_func.py:19 imports
    assert callable(MSDA.ms_deform_attn_forward) and callable(MSDA.ms_deform_attn_backward)
    with pytest.raises(RuntimeError):                                          # CPU tensors: the reference's AT_ASSERTM(.is_cuda())
        v = torch.zeros(1, 4, 2, 8)
        MSDA.ms_deform_attn_forward(v, torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 1, 2, 1, 1, 2), torch.zeros(1, 1, 2, 1, 1), 64)


def test_pass_bounds_cover_the_chunk():
    """Frame passes: every frame exactly once, no pass longer than `fbatch`, tapered head / tail only when there is room."""
    for n in (1, 3, 17, 40, 41, 44, 63, 100, 120, 240):
        for fb in (8, 20, 40):
            for taper in (False, True):
                b = MDQE.pass_bounds(n, fb, taper)
                assert b[-1] == n and b == sorted(set(b)) and all(y - x <= fb for x, y in zip([0] + b, b)), (n, fb, taper, b)
    assert MDQE.pass_bounds(120, 40, True) == [20, 60, 100, 120]
    assert MDQE.pass_bounds(120, 40, False) == [40, 80, 120]
    assert MDQE.pass_bounds(120, 40, True, tail=8) == [20, 60, 100, 112, 120]
    assert MDQE.pass_bounds(30, 40, True) == [30]
    assert MDQE.pass_bounds(63, 40, True) == [20, 41, 63]                      # (not 20 / 40 / 3: no sliver of a last pass)
    assert MDQE.pass_bounds(61, 40, True) == [20, 40, 61]


def test_stacked_view_of_per_frame_tensors():
    """The mapper hands over a list of frames; when they are consecutive views of one block the upload copies chunks of it."""
    v = torch.arange(5 * 3 * 4 * 6, dtype=torch.uint8).reshape(5, 3, 4, 6)
    s = MDQE.stacked_view(list(v))
    assert s is not None and s.shape == v.shape and s.data_ptr() == v.data_ptr() and torch.equal(s, v)
    assert MDQE.stacked_view([v[0], v[2], v[3]]) is None                       # a gap
    assert MDQE.stacked_view([f.clone() for f in v]) is None                   # separate allocations
    assert MDQE.stacked_view([v[0]]) is None
    assert MDQE.stacked_view(list(v[:, :, :, :3])) is None                      # non-contiguous frames
    tail = list(v[3:])                                                         # a suffix of the block: still one view
    s = MDQE.stacked_view(tail)
    assert s is not None and torch.equal(s, v[3:])
