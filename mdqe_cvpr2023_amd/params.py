"""Parameter manifest of the eval graph under the reference's checkpoint names, and a seeded
reference-style random initialisation for benchmarking (no checkpoints are reachable offline).

Names/shapes follow the reference modules (prefix `detr.`, SURVEY.md §5 "Checkpoint / resume"):
mdqe/models/mdqe.py:31-45 (input_proj), transformer_enc.py:14-28,77-96 (encoder),
transformer_dec.py:37-64,281-334 (decoder + heads), segmentation.py:12-40,66-109 (mask head),
ops/modules/ms_deform_attn.py:70-76 (MSDeformAttn), detectron2 ResNet naming for the backbone.
Aliased entries of the reference checkpoint (transformer_dec.decoder.{bbox_embed,point2pos_proj,norm})
are accepted on load and ignored in favour of their primary names.
"""
import math
import os
from collections import OrderedDict

import torch

from .config import MDQEConfig

RESNET_BLOCKS = {"R50": (3, 4, 6, 3), "R101": (3, 4, 23, 3)}
ALIASES = {
    "detr.transformer_dec.decoder.point2pos_proj.": "detr.transformer_dec.point2pos_proj.",
    "detr.transformer_dec.decoder.bbox_embed.": "detr.transformer_dec.bbox_embed.",
    "detr.transformer_dec.decoder.norm.": "detr.transformer_dec.decoder_norm.",
}


def resnet_manifest(kind="R50", p="detr.backbone.0.backbone"):
    m = OrderedDict()

    def conv(name, cout, cin, k):
        m[f"{name}.weight"] = (cout, cin, k, k)
        for s in ("weight", "bias", "running_mean", "running_var"):
            m[f"{name}.norm.{s}"] = (cout,)

    conv(f"{p}.stem.conv1", 64, 3, 7)
    cin = 64
    for si, nb in enumerate(RESNET_BLOCKS[kind]):
        mid, cout = 64 * 2 ** si, 256 * 2 ** si
        for b in range(nb):
            q = f"{p}.res{si + 2}.{b}"
            if b == 0:
                conv(q + ".shortcut", cout, cin, 1)
            conv(q + ".conv1", mid, cin, 1)
            conv(q + ".conv2", mid, mid, 3)
            conv(q + ".conv3", cout, mid, 1)
            cin = cout
    return m


def swin_manifest(cfg: MDQEConfig, p="detr.backbone.0.backbone"):
    """SwinTransformerV2 parameters (mdqe/backbone/swin_transformer_v2.py:97-145,217-229,305-309,459-464,605-611)."""
    m = OrderedDict()
    C0 = cfg.swin_embed_dim
    m[f"{p}.patch_embed.proj.weight"] = (C0, 3, 4, 4)
    m[f"{p}.patch_embed.proj.bias"] = (C0,)
    m[f"{p}.patch_embed.norm.weight"] = (C0,)
    m[f"{p}.patch_embed.norm.bias"] = (C0,)
    nl = len(cfg.swin_depths)
    for i, depth in enumerate(cfg.swin_depths):
        dim, nh = C0 * 2 ** i, cfg.swin_heads[i]
        hid = int(dim * cfg.swin_mlp_ratio)
        for j in range(depth):
            q = f"{p}.layers.{i}.blocks.{j}"
            for n in ("norm1", "norm2"):
                m[f"{q}.{n}.weight"] = (dim,)
                m[f"{q}.{n}.bias"] = (dim,)
            m[q + ".attn.logit_scale"] = (nh, 1, 1)
            m[q + ".attn.cpb_mlp.0.weight"] = (512, 2)
            m[q + ".attn.cpb_mlp.0.bias"] = (512,)
            m[q + ".attn.cpb_mlp.2.weight"] = (nh, 512)
            m[q + ".attn.qkv.weight"] = (3 * dim, dim)
            m[q + ".attn.q_bias"] = (dim,)
            m[q + ".attn.v_bias"] = (dim,)
            m[q + ".attn.proj.weight"] = (dim, dim)
            m[q + ".attn.proj.bias"] = (dim,)
            m[q + ".mlp.fc1.weight"] = (hid, dim)
            m[q + ".mlp.fc1.bias"] = (hid,)
            m[q + ".mlp.fc2.weight"] = (dim, hid)
            m[q + ".mlp.fc2.bias"] = (dim,)
        if i < nl - 1:
            m[f"{p}.layers.{i}.downsample.reduction.weight"] = (2 * dim, 4 * dim)
            m[f"{p}.layers.{i}.downsample.norm.weight"] = (2 * dim,)
            m[f"{p}.layers.{i}.downsample.norm.bias"] = (2 * dim,)
    for i in (1, 2, 3):
        m[f"{p}.norm{i}.weight"] = (C0 * 2 ** i,)
        m[f"{p}.norm{i}.bias"] = (C0 * 2 ** i,)
    return m


def _lin(m, name, out_f, in_f):
    m[name + ".weight"] = (out_f, in_f)
    m[name + ".bias"] = (out_f,)


def _norm(m, name, c):
    m[name + ".weight"] = (c,)
    m[name + ".bias"] = (c,)


def _mlp(m, name, dims):
    for i in range(len(dims) - 1):
        _lin(m, f"{name}.layers.{i}", dims[i + 1], dims[i])


def _msda(m, name, C, nh, lvl, pts, pred_offsets):
    _lin(m, name + ".value_proj", C, C)
    _lin(m, name + ".output_proj", C, C)
    _lin(m, name + ".attention_weights", nh * lvl * pts, C)
    _lin(m, name + (".sampling_offsets" if pred_offsets else ".sampling_grid_offsets"), nh * lvl * pts * 2, C)


def head_manifest(cfg: MDQEConfig, p="detr"):
    """Everything except the backbone."""
    C, nh, F = cfg.hidden_dim, cfg.nheads, cfg.d_ffn
    m = OrderedDict()
    e = f"{p}.transformer_enc"
    m[e + ".level_embed"] = (cfg.n_levels, C)
    for i in range(cfg.enc_layers):
        q = f"{e}.encoder.layers.{i}"
        _msda(m, q + ".self_attn", C, nh, cfg.n_levels, cfg.enc_points, True)
        _norm(m, q + ".norm1", C)
        _lin(m, q + ".linear1", F, C)
        _lin(m, q + ".linear2", C, F)
        _norm(m, q + ".norm2", C)
    _norm(m, e + ".encoder.norm", C)
    d = f"{p}.transformer_dec"
    _norm(m, d + ".decoder_norm", C)
    _mlp(m, d + ".bbox_embed", (C, C, C, 4))
    _lin(m, d + ".point2pos_proj", C, 2)
    for i in range(cfg.dec_layers):
        q = f"{d}.decoder.layers.{i}"
        for sa in (".self_attn", ):
            m[q + sa + ".in_proj_weight"] = (3 * C, C)
            m[q + sa + ".in_proj_bias"] = (3 * C,)
            _lin(m, q + sa + ".out_proj", C, C)
        _norm(m, q + ".norm1", C)
        _msda(m, q + ".cross_attn", C, nh, cfg.n_levels, cfg.dec_points, False)
        _norm(m, q + ".norm2", C)
        _lin(m, q + ".linear1", F, C)
        _lin(m, q + ".linear2", C, F)
        _norm(m, q + ".norm3", C)
        _lin(m, q + ".time_weights", 1, C)
        m[q + ".self_attn_inst.in_proj_weight"] = (3 * C, C)
        m[q + ".self_attn_inst.in_proj_bias"] = (3 * C,)
        _lin(m, q + ".self_attn_inst.out_proj", C, C)
        _norm(m, q + ".norm1_inst", C)
        if cfg.dec_temporal:
            _msda(m, q + ".temp_attn_inst", C, nh, cfg.n_frames, cfg.dec_points, False)
        _norm(m, q + ".norm2_inst", C)
        _lin(m, q + ".linear1_inst", F, C)
        _lin(m, q + ".linear2_inst", C, F)
        _norm(m, q + ".norm3_inst", C)
    _mlp(m, d + ".rpn_cls_embed", (C, C, C, cfg.num_classes))
    _mlp(m, d + ".cls_embed", (C, C, C, cfg.num_classes))
    _mlp(m, d + ".track_embed", (C, C, C, cfg.query_embed_dim))
    h = d + ".mask_head"
    for i in (1, 2, 3):
        m[f"{h}.lay{i}.weight"] = (C, C, 3, 3)
        m[f"{h}.lay{i}.bias"] = (C,)
        _norm(m, f"{h}.gn{i}", C)
    for name, oc in (("out_lay1", C), ("out_lay2", cfg.mask_dim)):
        m[f"{h}.{name}.depthwise.weight"] = (C, 1, 5, 5)
        m[f"{h}.{name}.depthwise.bias"] = (C,)
        m[f"{h}.{name}.pointwise.weight"] = (oc, C, 1, 1)
        m[f"{h}.{name}.pointwise.bias"] = (oc,)
        _norm(m, f"{h}.{name}.gn", oc)
    m[h + ".out_uplay.weight"] = (C, 1, 1, 1)
    m[h + ".out_uplay.bias"] = (C,)
    for i in (1, 2):
        m[f"{h}.adapter{i}.weight"] = (C, C, 1, 1)
        m[f"{h}.adapter{i}.bias"] = (C,)
    _mlp(m, d + ".mask_embed", (C, C, C, cfg.mask_dim))
    nb = len(cfg.backbone_channels)
    for l in range(cfg.n_levels):
        if l < nb:
            m[f"{p}.input_proj.{l}.0.weight"] = (C, cfg.backbone_channels[l], 1, 1)
        else:
            cin = cfg.backbone_channels[-1] if l == nb else C
            m[f"{p}.input_proj.{l}.0.weight"] = (C, cin, 3, 3)
        m[f"{p}.input_proj.{l}.0.bias"] = (C,)
        _norm(m, f"{p}.input_proj.{l}.1", C)
    return m


def full_manifest(cfg: MDQEConfig):
    m = OrderedDict()
    if cfg.backbone in RESNET_BLOCKS:
        m.update(resnet_manifest(cfg.backbone))
    elif cfg.backbone == "SwinV2":
        m.update(swin_manifest(cfg))
    m.update(head_manifest(cfg))
    return m


def msda_dir_grid(n_heads, n_lvl, n_points, scale=8.0):
    """Fixed direction grid of MSDeformAttn._reset_parameters (ops/modules/ms_deform_attn.py:81-87): [H,L,K,2]."""
    th = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
    g = torch.stack([th.cos(), th.sin()], -1)
    g = g / g.abs().max(-1, keepdim=True)[0]
    g = g.view(n_heads, 1, 1, 2).repeat(1, n_lvl, n_points, 1)
    for k in range(n_points):
        g[:, :, k, :] *= k + 1
    return g / n_points * scale


def random_state(cfg: MDQEConfig, seed=0, remove_zero_init_trap=True):
    """Reference-style init on CPU: xavier-uniform matrices (transformer_dec.py:68-71; MSDeformAttn
    value/output proj :103-106), kaiming-uniform(a=1) mask-head convs (segmentation.py:37-40), MSRA-like
    backbone convs with unit FrozenBN, MSDeformAttn offsets bias = direction grid (:88-92).
    remove_zero_init_trap (BASELINE.md §3 / SURVEY.md §8d): seeded N(0,0.02) noise on the tensors the
    reference zero-initialises and cls biases 0 instead of -4.6, so the data-dependent paths do real work."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    C = cfg.hidden_dim
    for name, shape in full_manifest(cfg).items():
        if name.endswith("running_var"):
            t = torch.ones(shape)
        elif name.endswith("running_mean"):
            t = torch.zeros(shape)
        elif ".norm." in name and "backbone" in name:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        elif len(shape) == 1 and not name.endswith(("q_bias", "v_bias")):
            is_gamma = name.endswith(".weight")
            t = torch.ones(shape) if is_gamma else torch.zeros(shape)
        elif name.endswith("level_embed"):
            t = torch.randn(shape, generator=g)
        elif name.endswith("logit_scale"):
            t = torch.log(10 * torch.ones(shape))
        elif "backbone" in name and cfg.backbone == "SwinV2":
            t = torch.randn(shape, generator=g) * 0.02                 # trunc_normal_(std=.02) stand-in
        elif "backbone" in name:
            fan_out = shape[0] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        else:
            fan_in = shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1)
            fan_out = shape[0] * (shape[2] * shape[3] if len(shape) == 4 else 1)
            a = math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        sd[name] = t
    # MSDeformAttn specifics
    for name in list(sd):
        if name.endswith(".sampling_offsets.bias"):          # encoder
            grid = msda_dir_grid(cfg.nheads, cfg.n_levels, cfg.enc_points)
            grid = grid * 0.05 * torch.arange(1, cfg.n_levels + 1, dtype=torch.float32).view(1, -1, 1, 1)
            sd[name] = grid.reshape(-1).clone()
            sd[name.replace(".bias", ".weight")].zero_()
        if name.endswith(".sampling_grid_offsets.weight") or name.endswith(".attention_weights.weight"):
            sd[name].zero_()
    bias_value = -math.log((1 - 0.01) / 0.01)
    for h in ("cls_embed", "rpn_cls_embed"):
        sd[f"detr.transformer_dec.{h}.layers.2.bias"].fill_(bias_value)
    if remove_zero_init_trap:
        for name in list(sd):
            if name.endswith((".sampling_offsets.weight", ".sampling_grid_offsets.weight", ".attention_weights.weight",
                              ".sampling_grid_offsets.bias", ".attention_weights.bias")):
                sd[name] += torch.randn(sd[name].shape, generator=g) * 0.02
        for h in ("cls_embed", "rpn_cls_embed"):
            sd[f"detr.transformer_dec.{h}.layers.2.bias"].zero_()
        # class logits of an untrained head are all alike (spread 0.25): widen them and centre the threshold so that roughly a
        # quarter of the queries pass APPLY_CLS_THRES and 4-8 instances per clip survive NMS + rescoring (R50 360p, seed 0)
        sd["detr.transformer_dec.cls_embed.layers.2.weight"] *= float(os.environ.get("MDQE_SYNTH_CLS_GAIN", "8.0"))
        sd["detr.transformer_dec.cls_embed.layers.2.bias"].fill_(float(os.environ.get("MDQE_SYNTH_CLS_BIAS", "-14.5")))
        # Untrained attention averages over near-random values, which adds (almost) the same vector to every token in every
        # layer: after 6+6 layers all 196 queries are copies of each other (pairwise cosine 0.98), inference_clip's duplicate
        # removal and mask NMS leave ONE instance per clip and tracker / NMS / up-sampling idle.  Damping the residual-branch
        # output projections keeps the tokens location-specific, so the data-dependent stages see several instances per
        # clip, as on real OVIS video (synthetic weights only; any released checkpoint is loaded as is).
        k = float(os.environ.get("MDQE_SYNTH_RES_SCALE", "0.3"))
        for name in list(sd):
            if "backbone" in name:
                continue
            if name.endswith((".output_proj.weight", ".out_proj.weight", ".linear2.weight", ".linear2_inst.weight")):
                sd[name] *= k
    return sd
