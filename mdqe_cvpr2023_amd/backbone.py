"""`build_swinv2_backbone` for detectron2's BACKBONE_REGISTRY (mdqe/backbone/swin_transformer_v2.py:675-702).

The MI355X `MDQE` builds its backbone itself (engine.Engine.backbone_swin); this module serves callers that go through detectron2's
`build_backbone(cfg)`: a `SwinTransformerV2` with the reference's parameter names (`patch_embed.*`, `layers.{i}.blocks.{j}.*`,
`layers.{i}.downsample.*`, `norm{1,2,3}.*`), `forward(x) -> {"stage3", "stage4", "stage5"}` NCHW maps and `output_shape()`, computed by the
same HIP kernels (csrc/swin.hip + the fp32 MFMA GEMM).  Eval only."""
from collections import OrderedDict, namedtuple

import torch
import torch.nn as nn

from .config import MDQEConfig
from .params import swin_manifest

PREFIX = "detr.backbone.0.backbone"

try:
    from detectron2.layers import ShapeSpec
except Exception:                                     # detectron2 absent in this image
    ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=(None, None, None, None))


class SwinTransformerV2(nn.Module):
    """Drop-in for the object `build_swinv2_backbone` returns (swin_transformer_v2.py:485-673) on the eval path.
    forward(x): x [B, 3, H, W] fp32, already normalised and padded (what detectron2's meta-architectures hand a backbone;
    mdqe/mdqe.py:316-319 pads to multiples of 32, which this implementation requires)."""

    def __init__(self, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0,
                 out_features=("stage3", "stage4", "stage5"), state_dict=None, device="cuda", seed=0):
        super().__init__()
        self.cfg = MDQEConfig(backbone="SwinV2", swin_embed_dim=int(embed_dim), swin_depths=tuple(int(d) for d in depths),
                              swin_heads=tuple(int(h) for h in num_heads), swin_window=int(window_size), swin_mlp_ratio=float(mlp_ratio),
                              backbone_channels=tuple(int(embed_dim) * 2 ** i for i in (1, 2, 3)),
                              pixel_mean=(0.0, 0.0, 0.0), pixel_std=(1.0, 1.0, 1.0), device=str(device))
        self.device = torch.device(device)
        man = swin_manifest(self.cfg, p=PREFIX)
        self.out_features = [f for f in out_features]
        have = {"stage%d" % (i + 2) for i in (1, 2, 3)}
        if not set(self.out_features) <= have:
            raise ValueError("SwinTransformerV2 (MI355X eval path): out_features must be among %s (MODEL.SWIN.OUT_FEATURES of the shipped "
                             "configs), got %s" % (sorted(have), self.out_features))
        g = torch.Generator().manual_seed(seed)
        for name, shape in man.items():
            short = name[len(PREFIX) + 1:]
            if state_dict is not None:
                t = state_dict[short] if short in state_dict else state_dict[name]
            elif short.endswith("logit_scale"):
                t = torch.log(10 * torch.ones(shape))                         # swin_transformer_v2.py:99
            elif len(shape) == 1:                                             # norms: weight 1, bias 0; Linear biases 0
                t = torch.ones(shape) if short.endswith(".weight") else torch.zeros(shape)
            else:
                t = torch.randn(shape, generator=g) * 0.02                    # trunc_normal_(std=.02), :617-619
            self._reg(short, t.detach().clone().float())
        C0 = self.cfg.swin_embed_dim
        self._out_feature_channels = {"stage%d" % (i + 2): C0 * 2 ** i for i in range(len(depths))}
        self._out_feature_strides = {"stage%d" % (i + 2): 4 * 2 ** i for i in range(len(depths))}
        self._engine = None

    def _reg(self, dotted, tensor):
        parts = dotted.split(".")
        m = self
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, nn.Module())
            m = m._modules[p]
        m.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        for key in list(state_dict):                   # the reference's fixed buffers (swin_transformer_v2.py:120,133): accepted and dropped
            if key.startswith(prefix) and key.endswith(("relative_coords_table", "relative_position_index", "attn_mask")):
                state_dict.pop(key)
        super()._load_from_state_dict(state_dict, prefix, *a, **k)
        self._engine = None                            # repack on the next call

    def train(self, mode=True):
        """The reference's `train()` returns None (swin_transformer_v2.py:661-664), so nothing can rely on chaining; this one returns
        self and refuses training."""
        if mode:
            raise RuntimeError("mdqe_cvpr2023_amd.backbone.SwinTransformerV2 implements the eval-only path")
        return super().train(False)

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n]) for n in self.out_features}

    @property
    def size_divisibility(self):
        return 32

    @property
    def engine(self):
        if self._engine is None:
            from .engine import Engine
            sd = OrderedDict((PREFIX + "." + k, v) for k, v in self.state_dict().items())
            with torch.cuda.device(self.device):
                self._engine = Engine(self.cfg, sd, self.device, only_backbone=True)
        return self._engine

    @torch.no_grad()
    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3 or not x.is_cuda:
            raise RuntimeError("SwinTransformerV2: expected a CUDA tensor [B, 3, H, W]")
        H, W = int(x.shape[-2]), int(x.shape[-1])
        if H % 32 or W % 32:
            raise RuntimeError("SwinTransformerV2 (MI355X): H and W must be multiples of 32 (the meta-architecture pads to "
                               "size_divisibility 32, mdqe/mdqe.py:65,318), got %dx%d" % (H, W))
        with torch.cuda.device(x.device), torch.autocast(device_type="cuda", enabled=False):
            eng = self.engine
            geo = eng.geometry(H, W)
            outs = eng.backbone_swin(x.float().contiguous(), geo)               # [stage3, stage4, stage5] NHWC
        res = {}
        for i, name in enumerate(("stage3", "stage4", "stage5")):
            if name in self.out_features:
                res[name] = outs[i].permute(0, 3, 1, 2)                         # NCHW view, as the reference returns (:655-657)
        return res


def build_swinv2_backbone(cfg, input_shape=None):
    """Same signature and config keys as the reference's builder (swin_transformer_v2.py:675-702); `input_shape.channels` must be 3."""
    if input_shape is not None and getattr(input_shape, "channels", 3) not in (3, None):
        raise ValueError("build_swinv2_backbone: 3 input channels expected")
    sw = cfg.MODEL.SWIN
    return SwinTransformerV2(embed_dim=sw.EMBED_DIM, depths=sw.DEPTHS, num_heads=sw.NUM_HEADS, window_size=sw.WINDOW_SIZE,
                             mlp_ratio=sw.MLP_RATIO, out_features=sw.OUT_FEATURES, device=str(cfg.MODEL.DEVICE))


_STATE = {"state": "detectron2 not imported"}


def register_backbone_with_detectron2(registry=None, takeover=None):
    """BACKBONE_REGISTRY["build_swinv2_backbone"] -> this module's builder, under the same policy as the meta-architecture
    (meta_arch.register_with_detectron2): the reference's function, if registered, stays available as
    "build_swinv2_backbone_reference"; `MDQE_MI355X_REGISTER=alias` only adds "build_swinv2_backbone_mi355x"."""
    import logging
    import os
    log = logging.getLogger("mdqe_cvpr2023_amd")
    if registry is None:
        try:
            from detectron2.modeling import BACKBONE_REGISTRY as registry
        except (ImportError, OSError) as e:
            _STATE.update(state="detectron2 not importable", error="%s: %s" % (type(e).__name__, e))
            return dict(_STATE)
    if takeover is None:
        takeover = os.environ.get("MDQE_MI355X_REGISTER", "replace") != "alias"
    objs = registry._obj_map

    def build_swinv2_backbone_mi355x(cfg, input_shape=None):
        return build_swinv2_backbone(cfg, input_shape)
    if "build_swinv2_backbone_mi355x" not in objs:
        registry.register(build_swinv2_backbone_mi355x)
    done = {"state": "alias only"}
    if takeover:
        name = "build_swinv2_backbone"
        prev = objs.get(name)
        if prev is None:
            registry._do_register(name, build_swinv2_backbone)
        elif prev is not build_swinv2_backbone:
            objs[name + "_reference"] = prev
            objs[name] = build_swinv2_backbone
            log.warning("BACKBONE_REGISTRY['%s'] now builds the MI355X SwinTransformerV2; the function registered before stays selectable as "
                        "'%s_reference'", name, name)
        if not getattr(registry, "_mdqe_mi355x_backbone_guard", False):
            inner = registry._do_register

            def _do_register(nm, obj, _inner=inner, _objs=objs):
                if nm == name and _objs.get(name) is build_swinv2_backbone and obj is not build_swinv2_backbone:
                    _objs[name + "_reference"] = obj
                    log.warning("a second '%s' was registered after mdqe_cvpr2023_amd's: kept as '%s_reference'", name, name)
                    return
                _inner(nm, obj)
            registry._do_register = _do_register
            registry._mdqe_mi355x_backbone_guard = True
        done["state"] = "build_swinv2_backbone taken over"
    _STATE.clear()
    _STATE.update(done)
    return dict(done)
