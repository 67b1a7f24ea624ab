"""CPU ORACLE for the MDQE eval-only hot path  --  TEST INFRASTRUCTURE, NOT THE PRODUCT.

A from-scratch restatement (plain torch fp32 on CPU, functional, no nn.Module, no reference
imports) of what the reference computes on SURVEY.md §8 rows a1-a19.  Every function cites the
reference file:line it follows (paths relative to /root/reference).  It operates on a flat
``state`` dict whose keys are the reference's checkpoint names (prefix ``detr.``), so the same
weights drive the reference (in this container, via oracle/refshim.py), this oracle, and the HIP
product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package ``mdqe_cvpr2023_amd`` never does.

Parity status: PINNED against (i) the reference's own known-answer recipe for the native op
(mdqe/models/ops/test.py:21-60) and (ii) outputs of the reference python itself executed in the
build container (fixtures under tests/golden/, generator oracle/make_golden.py).  The ResNet-50
backbone (detectron2, third-party, un-vendored, version unpinned -- INSTALL.md:32-36) is restated
from the public definition and is "parity unpinned" (structure/shape checks only).
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# hyper-parameters (configs/R50_coco.yaml, configs/R50_ovis_360.yaml, mdqe/config.py:5-85)
# --------------------------------------------------------------------------------------------
@dataclass
class Hyper:
    hidden_dim: int = 256
    nheads: int = 8
    enc_layers: int = 6
    dec_layers: int = 6
    n_levels: int = 4
    enc_points: int = 4
    dec_points: int = 4
    n_frames: int = 4                 # INPUT.SAMPLING_FRAME_NUM
    num_classes: int = 25
    num_queries: int = 200            # rounded to a square, mdqe/mdqe.py:77-78
    query_embed_dim: int = 64
    window_inter_frame_asso: float = 5
    mlp_ratio: float = 4
    dec_temporal: bool = True
    # eval (mdqe/mdqe.py:183-192)
    clip_stride: int = 1
    n_frames_test: int = 4
    n_frames_window_test: int = 30
    n_max_inst: int = 120
    apply_cls_thres: float = 0.1
    detections_per_image: int = 15
    match_stride: int = 4
    size_divisibility: int = 32
    pixel_mean: Tuple[float, ...] = (123.675, 116.280, 103.530)
    pixel_std: Tuple[float, ...] = (58.395, 57.120, 57.375)
    backbone_strides: Tuple[int, ...] = (8, 16, 32)

    @property
    def n_query(self):
        return int(math.sqrt(self.num_queries)) ** 2

    @property
    def n_bins(self):
        return int(math.sqrt(self.n_query))

    @property
    def mask_dim(self):
        return self.hidden_dim // 8


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _mlp(sd, p, x, n):
    """MLP with exact GELU between layers (mdqe/models/misc.py:6-18)."""
    for i in range(n):
        x = _lin(sd, f"{p}.layers.{i}", x)
        if i < n - 1:
            x = F.gelu(x)
    return x


def inverse_sigmoid(x, eps=1e-5):
    """mdqe/util/misc.py:478-482."""
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def box_cxcywh_to_xyxy(b):
    """mdqe/util/box_ops.py:8-12."""
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)


def box_xyxy_to_cxcywh(b):
    """mdqe/util/box_ops.py:15-19."""
    x0, y0, x1, y1 = b.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], -1)


# --------------------------------------------------------------------------------------------
# a9: the native op.  Follows the CUDA kernel's arithmetic
# (mdqe/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84, 237-299), NOT grid_sample:
#   h_im = loc_y*H - 0.5 ; sample skipped unless -1 < h_im < H and -1 < w_im < W ;
#   four corners, each zero outside the map ; weights hh*hw, hh*lw, lh*hw, lh*lw.
# --------------------------------------------------------------------------------------------
def msda_forward(value: Tensor, shapes: Sequence[Tuple[int, int]], level_start: Sequence[int],
                 loc: Tensor, attn: Tensor) -> Tensor:
    """value [B,S,M,D]; loc [B,Q,M,L,P,2] (x,y); attn [B,Q,M,L,P] -> [B,Q,M*D]."""
    B, S, M, D = value.shape
    _, Q, _, L, P, _ = loc.shape
    out = value.new_zeros(B, Q, M, D)
    bi = torch.arange(B).view(B, 1, 1, 1)
    mi = torch.arange(M).view(1, 1, M, 1)
    for l in range(L):
        H, W = int(shapes[l][0]), int(shapes[l][1])
        st = int(level_start[l])
        w_im = loc[:, :, :, l, :, 0] * W - 0.5          # [B,Q,M,P]
        h_im = loc[:, :, :, l, :, 1] * H - 0.5
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h_low = torch.floor(h_im)
        w_low = torch.floor(w_im)
        lh, lw = h_im - h_low, w_im - w_low
        hh, hw = 1 - lh, 1 - lw
        h_low, w_low = h_low.long(), w_low.long()
        acc = value.new_zeros(B, Q, M, P, D)
        for dy, dx, wgt in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
            yy, xx = h_low + dy, w_low + dx
            ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
            idx = st + yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)     # [B,Q,M,P]
            v = value[bi, idx, mi]                                      # [B,Q,M,P,D]
            acc = acc + v * (wgt * ok)[..., None]
        out = out + (acc * attn[:, :, :, l, :, None]).sum(3)
    return out.reshape(B, Q, M * D)


def msda_backward(value: Tensor, shapes, level_start, loc: Tensor, attn: Tensor, grad_out: Tensor):
    """Gradients of msda_forward w.r.t. (value, loc, attn) -- what ms_deform_attn_cuda_backward returns
    (mdqe/models/ops/src/cuda/ms_deform_attn_cuda.cu:83-153; col2im kernels ms_deform_im2col_cuda.cuh:87-234).  The forward
    above is differentiable as written (floor() carries no gradient, exactly like the kernel's h_low / w_low), so autograd
    through it IS the restatement; the reference's own gradcheck (ops/test.py:63-86) pins its CUDA kernel the same way."""
    with torch.enable_grad():
        v, lo, at = (t.detach().clone().requires_grad_(True) for t in (value, loc, attn))
        out = msda_forward(v, shapes, level_start, lo, at)
        return list(torch.autograd.grad(out, (v, lo, at), grad_out))


# --------------------------------------------------------------------------------------------
# a8 / a13: MSDeformAttn module (mdqe/models/ops/modules/ms_deform_attn.py)
# --------------------------------------------------------------------------------------------
def msda_dir_grid(n_heads, n_lvl, n_points, scale=8.0):
    """Fixed direction grid, ms_deform_attn.py:81-87 -> [H, L, K, 2]."""
    th = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
    g = torch.stack([th.cos(), th.sin()], -1)
    g = g / g.abs().max(-1, keepdim=True)[0]
    g = g.view(n_heads, 1, 1, 2).repeat(1, n_lvl, n_points, 1)
    for k in range(n_points):
        g[:, :, k, :] *= k + 1
    return g / n_points * scale


def _msda_loc_weights(sd, p, query, ref_boxes, n_heads, n_lvl, n_points, pred_offsets):
    """sampling locations + softmax weights (ms_deform_attn.py:141-161 / 198-217)."""
    scale = 8.0
    lead = query.shape[:-1]
    ref = ref_boxes.reshape(*lead, 1, 1, 1, ref_boxes.shape[-1])
    if pred_offsets:
        off = _lin(sd, p + ".sampling_offsets", query).reshape(*lead, n_heads, n_lvl, n_points, 2)
    else:
        wh = ref[..., 2:]
        grid = sd[p + ".sampling_offsets"] if (p + ".sampling_offsets") in sd else msda_dir_grid(n_heads, n_lvl, n_points)
        base = grid.reshape(1, 1, n_heads, n_lvl, n_points, 2) * 0.5 * wh
        d = _lin(sd, p + ".sampling_grid_offsets", query).reshape(*lead, n_heads, n_lvl, n_points, 2)
        d = torch.where(d > -wh * scale, d, -wh * scale)
        d = torch.where(d < wh * scale, d, wh * scale)
        off = base + d
    loc = ref[..., :2] + off / scale
    aw = _lin(sd, p + ".attention_weights", query).reshape(*lead, n_heads, n_lvl * n_points)
    aw = F.softmax(aw, -1).reshape(*lead, n_heads, n_lvl, n_points)
    return loc, aw


def msda_spatial(sd, p, query, ref_boxes, x, shapes, mask, n_heads, n_points, pred_offsets):
    """MSDeformAttn.spatial_forward, ms_deform_attn.py:118-173."""
    B, N, C = x.shape
    L = len(shapes)
    value = _lin(sd, p + ".value_proj", x)
    if mask is not None:
        value = value.masked_fill(mask[..., None], 0.0)
    value = value.view(B, N, n_heads, C // n_heads)
    loc, aw = _msda_loc_weights(sd, p, query, ref_boxes, n_heads, L, n_points, pred_offsets)
    starts = np.concatenate([[0], np.cumsum([h * w for h, w in shapes])[:-1]]).tolist()
    out = msda_forward(value, shapes, starts, loc, aw)
    return _lin(sd, p + ".output_proj", out)


def msda_temporal(sd, p, query, ref_boxes, x, shapes, mask, n_heads, n_points, n_frames_cfg):
    """MSDeformAttn.temporal_clip_forward, ms_deform_attn.py:175-238.
    x: [B,T,N,C] with T == n_frames_cfg (caller pads, transformer_dec.py:382-386)."""
    B, T, N, C = x.shape
    value = _lin(sd, p + ".value_proj", x)
    if mask is not None:
        value = value.masked_fill(mask[..., None], 0.0)
    value = value.view(B, T, N, n_heads, C // n_heads)
    loc, aw = _msda_loc_weights(sd, p, query, ref_boxes, n_heads, n_frames_cfg, n_points, False)
    res = []
    st = 0
    for (H, W) in shapes:
        v_l = value[:, :, st:st + H * W].reshape(B, T * H * W, n_heads, C // n_heads)
        st += H * W
        res.append(msda_forward(v_l, [(H, W)] * n_frames_cfg, [t * H * W for t in range(n_frames_cfg)], loc, aw))
    return _lin(sd, p + ".output_proj", torch.stack(res).mean(0))


# --------------------------------------------------------------------------------------------
# a5: sine position embedding (mdqe/models/position_encoding.py:28-48)
# --------------------------------------------------------------------------------------------
def pos_sine(mask: Tensor, num_pos_feats: int, temperature=10000.0) -> Tensor:
    """mask [B,H,W] bool (True = padding) -> [B, 2*num_pos_feats, H, W]."""
    nm = ~mask
    y = nm.cumsum(1, dtype=torch.float32)
    x = nm.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    d = torch.arange(num_pos_feats, dtype=torch.float32)
    d = temperature ** (2 * torch.div(d, 2, rounding_mode="trunc") / num_pos_feats)
    px = x[..., None] / d
    py = y[..., None] / d
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), 4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), 4).flatten(3)
    return torch.cat((py, px), 3).permute(0, 3, 1, 2)


def padding_masks(n, feat_hw: Sequence[Tuple[int, int]], strides, image_sizes):
    """MaskedBackbone.mask_out_padding, mdqe/mdqe.py:44-57."""
    out = []
    for (H, W), s in zip(feat_hw, strides):
        m = torch.ones(n, H, W, dtype=torch.bool)
        for i, (h, w) in enumerate(image_sizes):
            m[i, : int(np.ceil(float(h) / s)), : int(np.ceil(float(w) / s))] = False
        out.append(m)
    return out


def make_reference_points(H, W):
    """mdqe/models/misc.py:21-29 (pixel centres, x then y)."""
    ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
    return torch.stack((rx.reshape(-1) / max(W, 1), ry.reshape(-1) / max(H, 1)), -1)


# --------------------------------------------------------------------------------------------
# a4: ResNet-50 (detectron2 build_resnet_backbone; third-party, parity UNPINNED)
# names follow d2's checkpoint: stem.conv1.{weight,norm.*}, res{2..5}.{i}.{shortcut,conv1,conv2,conv3}
# --------------------------------------------------------------------------------------------
RESNET_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


def _frozen_bn(sd, p, x, eps=1e-5):
    scale = sd[p + ".weight"] * (sd[p + ".running_var"] + eps).rsqrt()
    bias = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def _conv_bn(sd, p, x, stride=1, padding=0, relu=True):
    x = F.conv2d(x, sd[p + ".weight"], None, stride, padding)
    x = _frozen_bn(sd, p + ".norm", x)
    return F.relu(x) if relu else x


def resnet(sd, p, x, depth=50):
    """STRIDE_IN_1X1 False (configs/R50_coco.yaml:7-10): stride sits on the 3x3. Returns res3,res4,res5.
    detectron2 is absent here; pinned against transformers.ResNetModel (same architecture) in tests/test_resnet_pin_cpu.py."""
    x = _conv_bn(sd, p + ".stem.conv1", x, 2, 3)
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for si, nb in enumerate(RESNET_BLOCKS[depth]):
        for b in range(nb):
            q = f"{p}.res{si + 2}.{b}"
            s = 2 if (b == 0 and si > 0) else 1
            sc = _conv_bn(sd, q + ".shortcut", x, s, 0, relu=False) if (q + ".shortcut.weight") in sd else x
            y = _conv_bn(sd, q + ".conv1", x, 1, 0)
            y = _conv_bn(sd, q + ".conv2", y, s, 1)
            y = _conv_bn(sd, q + ".conv3", y, 1, 0, relu=False)
            x = F.relu(y + sc)
        if si >= 1:
            outs.append(x)
    return outs


# --------------------------------------------------------------------------------------------
# a6 + a7: input_proj + deformable encoder
# (mdqe/models/mdqe.py:79-105, transformer_enc.py:30-59,100-110,121-136)
# --------------------------------------------------------------------------------------------
def input_proj_and_flatten(sd, hp: Hyper, feats: List[Tensor], masks: List[Tensor], p="detr"):
    srcs, ms, poses = [], [], []
    npf = hp.hidden_dim // 2
    for l in range(hp.n_levels):
        if l < len(feats):
            s = F.conv2d(feats[l], sd[f"{p}.input_proj.{l}.0.weight"], sd[f"{p}.input_proj.{l}.0.bias"])
            m = masks[l]
        else:
            src_in = feats[-1] if l == len(feats) else srcs[-1]
            s = F.conv2d(src_in, sd[f"{p}.input_proj.{l}.0.weight"], sd[f"{p}.input_proj.{l}.0.bias"], 2, 1)
            m = F.interpolate(masks[len(feats) - 1][None].float(), size=s.shape[-2:]).to(torch.bool)[0]
        s = F.group_norm(s, 32, sd[f"{p}.input_proj.{l}.1.weight"], sd[f"{p}.input_proj.{l}.1.bias"], 1e-5)
        srcs.append(s)
        ms.append(m)
        poses.append(pos_sine(m, npf))
    shapes = [tuple(s.shape[-2:]) for s in srcs]
    x = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    mask = torch.cat([m.flatten(1) for m in ms], 1)
    lvl = sd[f"{p}.transformer_enc.level_embed"]
    pos = torch.cat([q.flatten(2).transpose(1, 2) + lvl[i].view(1, 1, -1) for i, q in enumerate(poses)], 1)
    return x, mask, pos, shapes


def encoder(sd, hp: Hyper, x, mask, pos, shapes, p="detr.transformer_enc", collect=None):
    BT = x.shape[0]
    ref = torch.cat([make_reference_points(h, w) for h, w in shapes])[None].expand(BT, -1, -1)
    boxes = torch.cat([ref, torch.ones_like(ref) * 0.1], -1)
    for i in range(hp.enc_layers):
        q = f"{p}.encoder.layers.{i}"
        x2 = msda_spatial(sd, q + ".self_attn", x + pos, boxes, x, shapes, mask, hp.nheads, hp.enc_points, True)
        x = _ln(sd, q + ".norm1", x + x2)
        x2 = _lin(sd, q + ".linear2", F.gelu(_lin(sd, q + ".linear1", x)))
        x = _ln(sd, q + ".norm2", x + x2)
        if collect is not None:
            collect.append(x)
    return _ln(sd, f"{p}.encoder.norm", x)


# --------------------------------------------------------------------------------------------
# a10: mask-feature head (mdqe/models/segmentation.py:42-63, 111-113)
# --------------------------------------------------------------------------------------------
def _dwsep(sd, p, x, relu=True):
    C = x.shape[1]
    x = F.conv2d(x, sd[p + ".depthwise.weight"], sd[p + ".depthwise.bias"], padding=2, groups=C)
    x = F.conv2d(x, sd[p + ".pointwise.weight"], sd[p + ".pointwise.bias"])
    oc = x.shape[1]
    x = F.group_norm(x, 32 if oc % 32 == 0 else 24, sd[p + ".gn.weight"], sd[p + ".gn.bias"], 1e-5)
    return F.relu(x) if relu else x


def mask_head(sd, enc: Tensor, shapes, p="detr.transformer_dec.mask_head") -> Tensor:
    """enc [BT,N,C] -> mask features [M, BT, H/4, W/4]  (mdqe/models/mdqe.py:107-117)."""
    lv, st = [], 0
    for (H, W) in shapes:
        lv.append(enc[:, st:st + H * W].transpose(1, 2).reshape(enc.shape[0], -1, H, W))
        st += H * W
    x, fpns = lv[2], [lv[1], lv[0]]
    for i, f in enumerate([None] + fpns):
        if f is not None:
            cur = F.conv2d(f, sd[f"{p}.adapter{i}.weight"], sd[f"{p}.adapter{i}.bias"])
            x = cur + F.interpolate(x, size=cur.shape[-2:], mode="nearest")
        x = F.conv2d(x, sd[f"{p}.lay{i + 1}.weight"], sd[f"{p}.lay{i + 1}.bias"], padding=1)
        x = F.gelu(F.group_norm(x, 8, sd[f"{p}.gn{i + 1}.weight"], sd[f"{p}.gn{i + 1}.bias"], 1e-5))
    x = _dwsep(sd, p + ".out_lay1", x)
    x = F.conv_transpose2d(x, sd[p + ".out_uplay.weight"], sd[p + ".out_uplay.bias"], stride=2,
                           output_padding=1, groups=x.shape[1])
    x = _dwsep(sd, p + ".out_lay2", x)
    return x.permute(1, 0, 2, 3).contiguous()      # '(B T) M H W -> B M T H W' with B=1, squeezed


# --------------------------------------------------------------------------------------------
# a11: query initialisation (transformer_dec.py:81-206)
# --------------------------------------------------------------------------------------------
def grid_guided_query_selection(cls_conf: Tensor, n_bins: int, return_score=False):
    """cls_conf [T,H,W,K] -> coords [T,Q,2] (x,y).  transformer_dec.py:81-109."""
    T, H, W, K = cls_conf.shape
    s = cls_conf.float().sigmoid().max(-1)[0].unsqueeze(1)
    H_up = (2 * H // n_bins + 1) * n_bins
    W_up = (2 * W // n_bins + 1) * n_bins
    s = F.interpolate(s, size=(H_up, W_up), mode="bilinear")
    r, t = H_up // n_bins, W_up // n_bins
    cells = s.view(T, n_bins, r, n_bins, t).permute(0, 1, 3, 2, 4).reshape(T, n_bins * n_bins, r * t)
    sel = cells.argmax(-1)                                       # first max wins
    gy = torch.arange(n_bins).view(1, n_bins, 1).expand(T, n_bins, n_bins).reshape(T, -1)
    gx = torch.arange(n_bins).view(1, 1, n_bins).expand(T, n_bins, n_bins).reshape(T, -1)
    row = gy * r + torch.div(sel, t, rounding_mode="floor")
    col = gx * t + sel % t
    idx = row * W_up + col
    qx = torch.fmod(idx, W_up) / W_up
    qy = (idx / W_up) / H_up                                     # TRUE division, :105-106
    coords = torch.stack([qx, qy], -1)
    return (coords, s) if return_score else coords


def inter_frame_query_association(q: Tensor, coords: Tensor, emb: Tensor, relpos: Tensor, window: float):
    """transformer_dec.py:111-145 (eval: w = window/2)."""
    T = q.shape[0]
    if T == 1:
        return q, coords, None
    ct = int((T - 1) / 2)
    w = window / 2
    sim = torch.einsum("tqc,kc->tqk", emb, emb[ct])
    idx = []
    for t in range(T):
        itv = max(t - ct, ct - t)
        m = (relpos > w * itv).any(-1)
        idx.append(sim[t].masked_fill(m, float("-inf")).softmax(-2).argmax(-2))
    idx = torch.stack(idx)                                       # [T,K]
    ar = torch.arange(T)[:, None]
    return q[ar, idx], coords[ar, idx], idx


def query_relpos_grid(n_bins):
    """transformer_dec.py:61-64."""
    i, j = torch.meshgrid(torch.arange(n_bins), torch.arange(n_bins), indexing="ij")
    ind = torch.stack([j, i], -1).view(-1, 2)
    return (ind[:, None] - ind[None]).abs()


def query_initialization(sd, hp: Hyper, enc: Tensor, shapes, p="detr.transformer_dec", dbg=None):
    T = enc.shape[0]
    H, W = shapes[0]
    starts = np.concatenate([[0], np.cumsum([h * w for h, w in shapes])]).tolist()
    conf = _mlp(sd, p + ".rpn_cls_embed", enc[:, :H * W], 3).view(T, H, W, -1)
    coords, score = grid_guided_query_selection(conf, hp.n_bins, True)
    grid = 2 * coords.view(T, hp.n_bins, hp.n_bins, 2) - 1
    qi = []
    for l, (Hl, Wl) in enumerate(shapes):
        f = enc[:, starts[l]:starts[l + 1]].transpose(1, 2).reshape(T, -1, Hl, Wl)
        qi.append(F.grid_sample(f, grid, mode="bilinear", padding_mode="border", align_corners=False))
    q = torch.stack(qi).mean(0).flatten(2).transpose(1, 2)       # [T,Q,C]
    emb = _mlp(sd, p + ".track_embed", q, 3)
    if dbg is not None:
        dbg.update(rpn_conf=conf, score_up=score, coords0=coords, content0=q, track_emb=emb)
    q, coords, idx = inter_frame_query_association(q, coords, emb, query_relpos_grid(hp.n_bins),
                                                   hp.window_inter_frame_asso)
    if dbg is not None:
        dbg.update(assoc_idx=idx)
    return q, coords


# --------------------------------------------------------------------------------------------
# a12-a14: decoder (transformer_dec.py:268-513, 208-265)
# --------------------------------------------------------------------------------------------
def _mha(sd, p, qk, v, n_heads):
    """nn.MultiheadAttention(batch_first) with q=k=qk, value=v, eval (transformer_dec.py:348-353)."""
    B, Q, C = qk.shape
    Wi, bi = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(qk, Wi[:C], bi[:C]).view(B, Q, n_heads, -1).transpose(1, 2)
    k = F.linear(qk, Wi[C:2 * C], bi[C:2 * C]).view(B, Q, n_heads, -1).transpose(1, 2)
    vv = F.linear(v, Wi[2 * C:], bi[2 * C:]).view(B, Q, n_heads, -1).transpose(1, 2)
    a = torch.softmax((q / math.sqrt(q.shape[-1])) @ k.transpose(-1, -2), -1)
    o = (a @ vv).transpose(1, 2).reshape(B, Q, C)
    return _lin(sd, p + ".out_proj", o)


def _clip_box(boxes, T, t0, t1):
    """Circumscribed clip box, transformer_dec.py:473-480.  boxes [T,Q,4] -> [1,Q,4]."""
    b = box_cxcywh_to_xyxy(boxes.transpose(0, 1)[None][:, :, t0:t1]).clamp(0, 1)
    b = torch.cat([b[..., :2].min(-2)[0], b[..., 2:].max(-2)[0]], -1)
    return box_xyxy_to_cxcywh(b)


def decoder(sd, hp: Hyper, x, coords, enc, shapes, mask, p="detr.transformer_dec", dbg=None):
    """DecoderDefAttn.forward + layers (eval, B=1).  x [T,Q,C], coords [T,Q,2], enc [T,N,C]."""
    T, Q, C = x.shape
    Tc = hp.n_frames
    ct = int((T - 1) / 2)
    nh = hp.nheads
    ref = torch.cat([coords, torch.ones_like(coords) * 0.1], -1)
    x_inst = x[ct][None]
    bb = lambda z: _mlp(sd, p + ".bbox_embed", _ln(sd, p + ".decoder_norm", z), 3)
    boxes = (bb(x) + inverse_sigmoid(ref)).sigmoid()
    x_pos = _lin(sd, p + ".point2pos_proj", boxes[..., :2])
    t0 = max(ct - int((Tc - 1) / 2), 0)
    t1 = ct + Tc
    ibox = _clip_box(boxes, T, t0, t1)
    ipos = _lin(sd, p + ".point2pos_proj", ibox[..., :2])
    # temporal frames (transformer_dec.py:368-386)
    itv = max(int(T / Tc), 1)
    ts = max(ct - int((Tc - 1) / 2) * itv, 0)
    tca = list(range(ts, T, itv))[:Tc]
    enc_t, mask_t = enc[tca][None], (mask[tca][None] if mask is not None else None)
    if enc_t.shape[1] < Tc:
        padn = Tc - enc_t.shape[1]
        enc_t = torch.cat([enc_t, enc_t[:, -1:].repeat(1, padn, 1, 1)], 1)
        if mask_t is not None:
            mask_t = torch.cat([mask_t, mask_t[:, -1:].repeat(1, padn, 1)], 1)
    for i in range(hp.dec_layers):
        q = f"{p}.decoder.layers.{i}"
        # box level: CA -> SA -> FFN  (transformer_dec.py:415-422)
        x2 = msda_spatial(sd, q + ".cross_attn", x + x_pos, boxes, enc, shapes, mask, nh, hp.dec_points, False)
        x = _ln(sd, q + ".norm2", x + x2)
        sx = x
        x = _ln(sd, q + ".norm1", x + _mha(sd, q + ".self_attn", x + x_pos, x, nh))
        x = _ln(sd, q + ".norm3", x + _lin(sd, q + ".linear2", F.gelu(_lin(sd, q + ".linear1", x))))
        sw = x
        # instance level (transformer_dec.py:361-409)
        tw = _lin(sd, q + ".time_weights", sw)[None]                      # [1,T,Q,1]
        fused = (F.softmax(tw, 1) * sx[None]).sum(1)                      # [1,Q,C]
        xi2 = fused
        if hp.dec_temporal:
            xi2 = msda_temporal(sd, q + ".temp_attn_inst", fused + ipos, ibox, enc_t, shapes, mask_t, nh,
                                hp.dec_points, Tc)
        x_inst = _ln(sd, q + ".norm2_inst", x_inst + xi2)
        x_inst = _ln(sd, q + ".norm1_inst", x_inst + _mha(sd, q + ".self_attn_inst", x_inst + ipos, x_inst, nh))
        x_inst = _ln(sd, q + ".norm3_inst",
                     x_inst + _lin(sd, q + ".linear2_inst", F.gelu(_lin(sd, q + ".linear1_inst", x_inst))))
        # iterative refinement (transformer_dec.py:492-503)
        boxes = (bb(x) + inverse_sigmoid(boxes)).sigmoid()
        x_pos = _lin(sd, p + ".point2pos_proj", boxes[..., :2])
        ibox = _clip_box(boxes, T, t0, t1)
        ipos = _lin(sd, p + ".point2pos_proj", ibox[..., :2])
        if dbg is not None:
            dbg.setdefault("x", []).append(x)
            dbg.setdefault("x_inst", []).append(x_inst)
            dbg.setdefault("boxes", []).append(boxes)
    return x, x_inst, boxes


def transformer_dec(sd, hp: Hyper, enc, mask, shapes, p="detr.transformer_dec", dbg=None):
    """Transformer_Dec.forward eval VIS branch (transformer_dec.py:208-265)."""
    q, coords = query_initialization(sd, hp, enc, shapes, p, dbg)
    if dbg is not None:
        dbg.update(query0=q, coords=coords)
    _, x_inst, _ = decoder(sd, hp, q, coords, enc, shapes, mask, p, dbg)
    n = _ln(sd, p + ".decoder_norm", x_inst)
    return {"cls": _mlp(sd, p + ".cls_embed", n, 3).sigmoid(),
            "mask_coeff": _mlp(sd, p + ".mask_embed", n, 3).tanh(),
            "query_embed": x_inst}


# --------------------------------------------------------------------------------------------
# a15: inference_clip (mdqe/mdqe.py:368-428)
# --------------------------------------------------------------------------------------------
def inference_clip(hp: Hyper, out: Dict[str, Tensor], mask_feats: Tensor):
    cls, coef, emb = out["cls"][0], out["mask_coeff"][0], out["query_embed"][0]
    thr = hp.apply_cls_thres
    ss, si = cls.max(-1)[0].sort(descending=True)
    valid = si[ss >= min(thr, float(ss[0]))]
    if valid.numel() > 1:
        e = F.normalize(emb[valid], dim=-1)
        ms = torch.triu(e @ e.t(), diagonal=1).max(0)[0]
        valid = valid[ms < 0.99][:10 * hp.detections_per_image]
    cls, coef, emb = cls[valid], coef[valid], emb[valid]
    mp = torch.einsum("qm,mthw->qthw", coef, mask_feats)
    nb = mp.gt(0.).flatten(1).sum(1) > 0
    cls, mp, emb = cls[nb], mp[nb], emb[nb]
    if cls.numel() > 0:
        mn = mp[:, ::2] if mp.shape[1] >= 5 else mp
        soft = F.interpolate(mn, scale_factor=0.5).flatten(1).sigmoid()
        hard = soft.gt(0.5).float()
        num = soft @ hard.t()
        den = soft.sum(-1)[:, None] + hard.sum(-1)[None] - num
        mi = torch.triu(num / (den + 1), diagonal=1).max(0)[0]
        cls = cls * (1 - mi[:, None])
        k = mi < 0.5
        cls, mp, emb = cls[k], mp[k], emb[k]
    soft = mp.sigmoid().flatten(1)
    hard = soft.gt(0.5).float()
    cls = cls * ((soft * hard).sum(1) / (hard.sum(1) + 1e-6))[:, None]
    sc, lab = cls.max(-1)
    order = sc.sort(descending=True)[1]
    n = max(int((sc > thr).sum()), 1)
    t = order[:n]
    return {"scores": sc[t], "pred_classes": lab[t], "cls_probs": cls[t], "pred_masks": mp[t],
            "query_embeds": emb[t]}


# --------------------------------------------------------------------------------------------
# a16: tracker (mdqe/tracking/OverTracker.py)
# --------------------------------------------------------------------------------------------
def ctt_similarity(saved, inp):
    """OverTracker.py:228-242 (bi-softmax)."""
    f = saved @ inp.t()
    Ns, Ni = f.shape
    Ws, Wi = (1 if Ns > 1 else 0), (1 if Ni > 1 else 0)
    d2t, t2d = f.softmax(0), f.softmax(1)
    if Ns == 1 and Ni == 1:
        return 0.5 * (d2t + t2d)
    return (Ws * d2t + Wi * t2d) / max(Ws + Wi, 1)


class Tracker:
    """Restatement of OverTracker (OverTracker.py:10-225); clip = dict from inference_clip + frame_idx."""

    def __init__(self, hp: Hyper, image_size):
        self.hp = hp
        self.T, self.win, self.stride = hp.n_frames_test, hp.n_frames_window_test, hp.clip_stride
        self.K, self.E = hp.num_classes, hp.hidden_dim
        self.size = tuple(image_size)
        self.max_inst = hp.n_max_inst
        self.num_inst = 0
        self.mem_len = self.win + self.T
        self.num_clips = self.win // self.stride + 2
        self.saved_idx = set()
        self.start_frame = 0
        self._init_memory(True)
        self.n_long = 15 // self.stride
        self.n_short = max(self.T, 5) // self.stride
        self.w_mem = torch.exp(torch.arange(self.n_long) * 0.25)
        self.untracked = torch.zeros(self.max_inst)
        self.embed_mem = torch.zeros(self.max_inst, self.E)

    def _init_memory(self, first=False):
        self.num_clip = 0 if first else 1
        self.start_frame = 0 if first else self.start_frame + self.win
        self.saved_idx.difference_update(range(self.start_frame))
        self.logits = torch.zeros(self.num_clips, self.max_inst, self.mem_len, *self.size)
        self.valid = torch.zeros(self.num_clips, self.max_inst, self.mem_len, dtype=torch.bool)
        self.cls = torch.zeros(self.num_clips, self.max_inst, self.K)
        self.embeds = torch.zeros(self.num_clips, self.max_inst, self.E)
        self.frame_idx = range(self.start_frame, self.start_frame + self.mem_len)

    def _update_memory(self, n_clip, r_idx, c_idx, clip):
        fi = clip["frame_idx"]
        s0 = max(min(fi) - self.start_frame, 0)
        s1 = max(fi) - self.start_frame
        a = fi.index(self.frame_idx[s0])
        b = fi.index(self.frame_idx[s1])
        self.logits[n_clip, r_idx, s0:s1 + 1] = clip["pred_masks"][c_idx, a:b + 1].float()
        self.valid[n_clip, r_idx, s0:s1 + 1] = True
        self.cls[n_clip, r_idx] = clip["cls_probs"][c_idx]
        self.embeds[n_clip, r_idx] = clip["query_embeds"][c_idx].float()
        self.untracked += 1
        self.untracked[r_idx] = 0
        if n_clip > 0:
            st = max(n_clip - 2, 0)
            qm = self.embeds[st:n_clip + 1][:, r_idx]
            w = self.w_mem[:qm.shape[0]].reshape(-1, 1, 1)
            vm = (qm != 0).any(-1)[..., None]
            self.embed_mem[r_idx] = (qm * w).sum(0) / (vm * w).sum(0).clamp(min=1)
        else:
            self.embed_mem[r_idx] = clip["query_embeds"][c_idx].float()

    @staticmethod
    def _siou(saved, inp):
        """OverTracker.py:92-113."""
        i = inp.flatten(1).gt(0.5).float()[None]
        s = saved.flatten(1).gt(0.5).float()[:, None]
        v = (s.any(-1) & i.any(-1)).unsqueeze(-1)
        num = s * i
        den = s + i - num
        return (num * v).sum(-1) / ((den * v).sum(-1) + 1e-6)

    def update(self, clip):
        from scipy.optimize import linear_sum_assignment
        n_in = len(clip["scores"])
        siou = None
        if self.num_inst == 0:
            mid = midx = list(range(n_in))
            self.num_inst += n_in
        else:
            qm = self.embed_mem[:self.num_inst]
            lo = (self.untracked[:self.num_inst] < self.n_long).nonzero().reshape(-1).tolist()
            sh = (self.untracked[:self.num_inst] < self.n_short).nonzero().reshape(-1).tolist()
            sm = torch.zeros(self.num_inst, n_in)
            sm[lo] = ctt_similarity(qm[lo], clip["query_embeds"])
            sm[sh] = 0.5 * (sm[sh] + ctt_similarity(qm[sh], clip["query_embeds"]))
            ii, si_ = [], []
            for o, f in enumerate(clip["frame_idx"]):
                if f in self.saved_idx and f >= self.start_frame:
                    ii.append(o)
                    si_.append(self.frame_idx.index(f))
            siou = torch.zeros(self.num_inst, n_in)
            if len(si_) > 0:
                im = clip["pred_masks"][:, ii].float()
                s = self.logits[:self.num_clip, :self.num_inst][:, :, si_]
                sv = self.valid[:self.num_clip, :self.num_inst].any(-1)
                s = s.sum(0) / sv.sum(0).clamp(min=1).reshape(-1, 1, 1, 1)
                siou = self._siou(s.sigmoid(), im.sigmoid())
            scores = siou + sm
            above = scores > 0.6
            scores = scores * above.float()
            r, c = linear_sum_assignment(scores.cpu(), maximize=True)
            mid, midx = [], []
            for ok, ri, ci in zip(above[r, c], r, c):
                if not ok:
                    continue
                midx.append(ci)
                mid.append(ri)
                siou[ri, ci] = -1
                sm[ri, ci] = 0
        un = [i for i in range(n_in) if i not in midx]
        rep = []
        for i in un:
            if siou[:, i].max() > 0.4 or sm[:, i].max() > 0.6:
                rep.append(i)
        un = [i for i in range(n_in) if i not in midx + rep and clip["scores"][i] > 2 * self.hp.apply_cls_thres]
        new = list(range(self.num_inst, self.num_inst + len(un)))
        mid, midx = list(mid) + new, list(midx) + un
        self._update_memory(self.num_clip, [int(v) for v in mid], [int(v) for v in midx], clip)
        self.saved_idx.update(clip["frame_idx"])
        self.num_clip += 1
        self.num_inst += len(new)

    def get_result(self, is_last=False):
        lg = self.logits[:self.num_clip, :self.num_inst]
        va = self.valid[:self.num_clip, :self.num_inst]
        cl = self.cls[:self.num_clip, :self.num_inst]
        qe = self.embeds[:self.num_clip, :self.num_inst]
        lg = lg.sum(0) / va.sum(0).clamp(min=1)[..., None, None]
        nv = max(self.saved_idx) - self.start_frame + 1
        ln = self.win if not is_last else int(nv)
        out_m = lg[:, :ln]
        vc = va.any(-1)[..., None]
        out_c = (cl * vc).sum(0) / vc.sum(0).clamp(min=1)
        nc = min(max(3, (self.T - 1) // self.stride), self.num_clip)
        qw = vc[-nc:] * self.w_mem[:nc].reshape(-1, 1, 1)
        oq = (qe[-nc:] * qw).sum(0) / qw.sum(0).clamp(min=1)
        if not is_last:
            n = self.num_inst
            self._init_memory(False)
            self.logits[0, :n, :self.mem_len - self.win] = lg[:n, self.win:]
            self.valid[0, :n, :self.mem_len - self.win] = va[:, :n, self.win:].any(0)
            self.cls[0, :n] = out_c
            self.embeds[0, :n] = oq
        return out_c, out_m


# --------------------------------------------------------------------------------------------
# a17 / a18: up-sampling and video merge
# --------------------------------------------------------------------------------------------
def aligned_bilinear(t: Tensor, factor: int) -> Tensor:
    """mdqe/util/misc.py:485-507, restated in closed form: output pixel (Y,X) reads the source at
    ((Y - f//2)/f, (X - f//2)/f) clamped to [0, h-1]x[0, w-1], bilinear."""
    if factor == 1:
        return t
    h, w = t.shape[-2:]
    f = factor

    def axis(n, size):
        o = torch.arange(f * size, dtype=torch.float32)
        src = ((o - f // 2).clamp(min=0)) * (float(size) / float(f * size))   # align_corners scale = h/(f*h)
        i0 = src.floor().long().clamp(max=size - 1)
        i1 = (i0 + 1).clamp(max=size - 1)
        return i0, i1, (src - i0.float())

    y0, y1, ly = axis(f, h)
    x0, x1, lx = axis(f, w)
    ly = ly.view(-1, 1)
    top = t[..., y0, :][..., x0] * (1 - lx) + t[..., y0, :][..., x1] * lx
    bot = t[..., y1, :][..., x0] * (1 - lx) + t[..., y1, :][..., x1] * lx
    return top * (1 - ly) + bot * ly


def inference_video(hp: Hyper, out_size, cls_clips: List[Tensor], mask_clips: List[Tensor]):
    """mdqe/mdqe.py:430-471."""
    total = cls_clips[-1].shape[0]
    cc = torch.stack([torch.cat([c, torch.zeros(total - c.shape[0], c.shape[1])]) for c in cls_clips])
    out_cls = 0.75 * cc.mean(0) + 0.25 * cc.max(0)[0]
    vids = []
    for i in range(total):
        vids.append(torch.cat([m[i] if i < m.shape[0] else torch.zeros_like(m[0]) for m in mask_clips], 0))
    labels = torch.arange(hp.num_classes).unsqueeze(0).repeat(out_cls.shape[0], 1).flatten()
    flat = out_cls.flatten()
    k = min(max(int(flat.gt(0.05).sum()), 10), flat.numel())     # clamp: the reference (:449-450) assumes >= 10 scores
    sc, ti = flat.topk(k, sorted=False)
    lab = labels[ti].tolist()
    inst = torch.div(ti, hp.num_classes, rounding_mode="floor")
    masks = [F.interpolate(vids[int(i)].unsqueeze(0), size=tuple(out_size), mode="nearest").squeeze(0) > 0.5
             for i in inst]
    return {"image_size": tuple(out_size), "pred_scores": sc.tolist(), "pred_labels": lab, "pred_masks": masks}


# --------------------------------------------------------------------------------------------
# a1-a3, a19: driver (mdqe/mdqe.py:291-366, 473-484)
# --------------------------------------------------------------------------------------------
def mask_bounding_boxes(masks: Tensor) -> Tensor:
    """detectron2 BitMasks.get_bounding_boxes (third-party, public algorithm; call sites mdqe/mdqe.py:526,554):
    [x_min, y_min, x_max + 1, y_max + 1] of the non-zero pixels of each [H,W] mask, zeros for an empty mask."""
    m = masks.to(torch.bool)
    out = torch.zeros(m.shape[0], 4)
    xa, ya = m.any(1), m.any(2)
    for i in range(m.shape[0]):
        x, y = torch.where(xa[i])[0], torch.where(ya[i])[0]
        if len(x) and len(y):
            out[i] = torch.tensor([float(x[0]), float(y[0]), float(x[-1] + 1), float(y[-1] + 1)])
    return out


def box_iou(a: Tensor, b: Tensor) -> Tensor:
    """mdqe/util/box_ops.py:30-43 (union clamped at 1e-3)."""
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    return inter / (area(a)[:, None] + area(b)[None] - inter).clamp(min=1e-3)


def inference_image(sd, hp: Hyper, frames: List[Tensor], backbone_fn, out_size=None, multi_cls=True):
    """COCO single-image branch: MDQE.forward (mdqe/mdqe.py:213-236) -> mdqe.forward (models/mdqe.py:62-70) -> decoder
    eval branch `is_coco` (transformer_dec.py:247-255) -> MDQE.inference_image (mdqe/mdqe.py:486-556).
    frames: the pseudo clip (hp.n_frames images).  Returns dict(cls, masks, scores, pred_classes, pred_masks, pred_boxes)."""
    video = preprocess(hp, frames)
    images, sizes = pad_frames(video, hp.size_divisibility)
    enc, mask, shapes, mf = frame_features(sd, hp, images, sizes, backbone_fn)        # mf [M,T,h,w]
    out = transformer_dec(sd, hp, enc, mask, shapes)
    cls = out["cls"][0]                                                                 # [Q,K] (sigmoid)
    masks = torch.einsum("qm,mthw->qthw", out["mask_coeff"][0], mf)                    # [Q,T,h,w]
    image_size = sizes[0]
    ct = int((hp.n_frames - 1) / 2)
    m = masks[:, ct]
    score = cls.max(-1)[0]
    idx = torch.nonzero(score >= min(hp.apply_cls_thres, float(score.max()))).reshape(-1)
    mc, m = cls[idx], m[idx]
    m = aligned_bilinear(m.unsqueeze(1), hp.match_stride).squeeze(1)[:, :image_size[0], :image_size[1]]
    soft = m.sigmoid()
    hard = soft > 0.5
    mc = mc * ((soft.flatten(1) * hard.flatten(1)).sum(1) / (hard.flatten(1).sum(1) + 1e-6))[:, None]
    if len(idx) > 0:                                                                    # box-IoU NMS, :519-531
        order = mc.max(-1)[0].sort(descending=True)[1]
        mc, m = mc[order], m[order]
        norm = torch.tensor([image_size[1], image_size[0], image_size[1], image_size[0]], dtype=torch.float32).reshape(1, -1)
        bx = mask_bounding_boxes(m.gt(0.)) / norm
        mc = mc * (1 - torch.triu(box_iou(bx, bx), diagonal=1).max(0)[0])[:, None]
    if multi_cls:
        ls = torch.nonzero(mc > hp.apply_cls_thres)
        ii, label = ls[:, 0], ls[:, 1]
        score, m = mc[ii, label], m[ii]
    else:
        score, label = mc.max(-1)
    oh, ow = out_size or image_size
    pm = F.interpolate(m.float().unsqueeze(1), size=[oh, ow], mode="bilinear").squeeze(1) > 0.
    return {"cls": cls, "masks": masks, "scores": score, "pred_classes": label, "pred_masks": pm,
            "pred_boxes": mask_bounding_boxes(pm)}


def preprocess(hp: Hyper, frames: List[Tensor]) -> List[Tensor]:
    mean = torch.tensor(hp.pixel_mean).view(3, 1, 1)
    std = torch.tensor(hp.pixel_std).view(3, 1, 1)
    return [(f.float() - mean) / std for f in frames]


def pad_frames(frames: List[Tensor], div: int) -> Tuple[Tensor, List[Tuple[int, int]]]:
    sizes = [tuple(f.shape[-2:]) for f in frames]
    H = (max(s[0] for s in sizes) + div - 1) // div * div
    W = (max(s[1] for s in sizes) + div - 1) // div * div
    out = torch.zeros(len(frames), frames[0].shape[0], H, W)
    for i, f in enumerate(frames):
        out[i, :, :f.shape[-2], :f.shape[-1]] = f
    return out, sizes


def frame_features(sd, hp: Hyper, images: Tensor, sizes, backbone_fn, p="detr"):
    """a3-a10 for a batch of frames: backbone -> input_proj -> encoder -> mask head."""
    feats = backbone_fn(images)
    masks = padding_masks(images.shape[0], [tuple(f.shape[-2:]) for f in feats], hp.backbone_strides, sizes)
    x, mask, pos, shapes = input_proj_and_flatten(sd, hp, feats, masks, p)
    enc = encoder(sd, hp, x, mask, pos, shapes, p + ".transformer_enc")
    mf = mask_head(sd, enc, shapes, p + ".transformer_dec.mask_head")
    return enc, mask, shapes, mf


def inference_vis(sd, hp: Hyper, frames: List[Tensor], backbone_fn, out_size=None, schedule="compute-once",
                  trace=None):
    """Driver.  schedule='as-reference' recomputes the window per clip (mdqe/mdqe.py:302,314 never
    updates window_end_idx); 'compute-once' runs each frame once -- identical results because every
    per-frame stage is frame-independent."""
    video = preprocess(hp, frames)
    L = len(video)
    img_size = tuple(video[0].shape[-2:])
    out_size = out_size or img_size
    cache = {}
    if schedule == "compute-once":
        images, sizes = pad_frames(video, hp.size_divisibility)
        enc_all, mask_all, shapes, mf_all = frame_features(sd, hp, images, sizes, backbone_fn)
    saved, last, tracker = 0, False, None
    cls_clips, mask_clips = [], []
    for start in range(0, L, hp.clip_stride):
        end = start + hp.n_frames_test
        if end > L:
            last, end = True, L
        if schedule == "as-reference":
            images, sizes = pad_frames(video[start:start + hp.n_frames_window_test], hp.size_divisibility)
            enc_w, mask_w, shapes, mf_w = frame_features(sd, hp, images, sizes, backbone_fn)
            idx = list(range(0, end - start))
            enc_c, mask_c, mf_c = enc_w[idx], mask_w[idx], mf_w[:, idx]
        else:
            idx = list(range(start, end))
            enc_c, mask_c, mf_c = enc_all[idx], mask_all[idx], mf_all[:, idx]
        out = transformer_dec(sd, hp, enc_c, mask_c, shapes)
        clip = inference_clip(hp, out, mf_c)
        clip["frame_idx"] = list(range(start, end))
        if trace is not None:
            trace.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in clip.items()})
        if tracker is None:
            tracker = Tracker(hp, mf_c.shape[-2:])
        tracker.update(clip)
        if last or (start + hp.clip_stride >= hp.n_frames_window_test * (saved + 1)):
            c, m = tracker.get_result(last)
            m = aligned_bilinear(m, hp.match_stride).sigmoid()[..., :img_size[0], :img_size[1]]
            cls_clips.append(c)
            mask_clips.append(m)
            saved += 1
        if last:
            break
    return inference_video(hp, out_size, cls_clips, mask_clips)


# --------------------------------------------------------------------------------------------
# a4': SwinV2 backbone (mdqe/backbone/swin_transformer_v2.py)
# --------------------------------------------------------------------------------------------
@dataclass
class SwinHyper:
    embed_dim: int = 192
    depths: Tuple[int, ...] = (2, 2, 18, 2)
    num_heads: Tuple[int, ...] = (6, 12, 24, 48)
    window_size: int = 12
    mlp_ratio: float = 4.0
    out_stages: Tuple[int, ...] = (1, 2, 3)        # stage3, stage4, stage5


def _window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def _window_reverse(w, ws, H, W):
    B = int(w.shape[0] / (H * W / ws / ws))
    x = w.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def swin_rel_tables(ws):
    """relative_coords_table [1,2ws-1,2ws-1,2] and relative_position_index [ws*ws, ws*ws] (WindowAttention.__init__,
    swin_transformer_v2.py:101-131, pretrained_window_size = 0)."""
    rh = torch.arange(-(ws - 1), ws, dtype=torch.float32)
    tab = torch.stack(torch.meshgrid([rh, rh], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
    tab = tab / (ws - 1) * 8
    tab = torch.sign(tab) * torch.log2(torch.abs(tab) + 1.0) / np.log2(8)
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return tab, rel.sum(-1)


def swin_attn_bias(sd, p, ws, nh):
    """16*sigmoid(cpb_mlp(table))[index] -> [nh, N, N] (swin_transformer_v2.py:164-169)."""
    tab, idx = swin_rel_tables(ws)
    t = F.linear(F.relu(F.linear(tab, sd[p + ".cpb_mlp.0.weight"], sd[p + ".cpb_mlp.0.bias"])), sd[p + ".cpb_mlp.2.weight"])
    b = t.view(-1, nh)[idx.view(-1)].view(ws * ws, ws * ws, nh).permute(2, 0, 1).contiguous()
    return 16 * torch.sigmoid(b)


def swin_shift_mask(H, W, ws):
    """BasicLayer.forward mask build (swin_transformer_v2.py:397-415)."""
    ss = ws // 2
    Hp, Wp = int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws
    img = torch.zeros(1, Hp, Wp, 1)
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
        for w in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = _window_partition(img, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def swin_block(sd, p, x, H, W, ws, shift, nh, mask):
    """SwinTransformerBlock.forward + WindowAttention.forward (swin_transformer_v2.py:147-186,236-290)."""
    B, L, C = x.shape
    shortcut = x
    x = x.view(B, H, W, C)
    pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pr, 0, pb))
    Hp, Wp = x.shape[1], x.shape[2]
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = _window_partition(x, ws).view(-1, ws * ws, C)
    a = p + ".attn"
    bias = torch.cat((sd[a + ".q_bias"], torch.zeros_like(sd[a + ".v_bias"]), sd[a + ".v_bias"]))
    qkv = F.linear(xw, sd[a + ".qkv.weight"], bias).reshape(xw.shape[0], ws * ws, 3, nh, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    attn = attn * torch.clamp(sd[a + ".logit_scale"], max=math.log(1. / 0.01)).exp()
    attn = attn + swin_attn_bias(sd, a, ws, nh).unsqueeze(0)
    if shift > 0:
        nW = mask.shape[0]
        attn = (attn.view(-1, nW, nh, ws * ws, ws * ws) + mask.unsqueeze(1).unsqueeze(0)).view(-1, nh, ws * ws, ws * ws)
    attn = attn.softmax(-1)
    o = (attn @ v).transpose(1, 2).reshape(xw.shape[0], ws * ws, C)
    o = _lin(sd, a + ".proj", o).view(-1, ws, ws, C)
    x = _window_reverse(o, ws, Hp, Wp)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x[:, :H, :W, :].contiguous().view(B, H * W, C)
    x = shortcut + _ln(sd, p + ".norm1", x)                                    # res-post-norm (:287-288)
    return x + _ln(sd, p + ".norm2", _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", x))))


def swin_patch_merge(sd, p, x, H, W):
    """PatchMerging.forward (swin_transformer_v2.py:311-335)."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 == 1 or W % 2 == 1:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    return _ln(sd, p + ".norm", F.linear(x, sd[p + ".reduction.weight"]))


def swinv2(sd, p, x, sh: SwinHyper):
    """SwinTransformerV2.forward (swin_transformer_v2.py:639-659): x [B,3,H,W] -> [stage3, stage4, stage5] NCHW."""
    if x.shape[3] % 4:
        x = F.pad(x, (0, 4 - x.shape[3] % 4))
    if x.shape[2] % 4:
        x = F.pad(x, (0, 0, 0, 4 - x.shape[2] % 4))
    x = F.conv2d(x, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=4)
    H, W = x.shape[2], x.shape[3]
    x = _ln(sd, p + ".patch_embed.norm", x.flatten(2).transpose(1, 2))
    outs = []
    nl = len(sh.depths)
    for i in range(nl):
        ws = sh.window_size // 2 if i == nl - 1 else sh.window_size
        mask = swin_shift_mask(H, W, ws)
        for j in range(sh.depths[i]):
            x = swin_block(sd, f"{p}.layers.{i}.blocks.{j}", x, H, W, ws, 0 if j % 2 == 0 else ws // 2, sh.num_heads[i], mask)
        if i in sh.out_stages:
            o = _ln(sd, f"{p}.norm{i}", x)
            outs.append(o.view(-1, H, W, o.shape[-1]).permute(0, 3, 1, 2).contiguous())
        if i < nl - 1:
            x = swin_patch_merge(sd, f"{p}.layers.{i}.downsample", x, H, W)
            H, W = (H + 1) // 2, (W + 1) // 2
    return outs
