"""Import shims for the *reference* python package (TEST INFRASTRUCTURE, this container only).

The reference (/root/reference, read-only) depends on detectron2 / torchvision / timm and on a
CUDA-only extension, none of which exist in this image.  This module registers minimal stand-in
*modules* (not reference code) so that the reference's own files can be imported from where they
lie and executed on CPU to produce golden vectors (see oracle/make_golden.py).

Nothing here ships to the GPU box as a dependency: tests/ only read the committed fixtures under
tests/golden/.  The reference has no CPU implementation of its native op
(mdqe/models/ops/src/cpu/ms_deform_attn_cpu.cpp:26,39 raise), so the stand-in
`MultiScaleDeformableAttention` module routes to the reference's *own* debug function
`ms_deform_attn_core_pytorch` (mdqe/models/ops/functions/ms_deform_attn_func.py:45-65).
"""
import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("MDQE_REFERENCE_ROOT", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


class _Registry:
    def __init__(self):
        self._d = {}

    def register(self, obj=None):
        def deco(o):
            self._d[o.__name__] = o
            return o
        return deco if obj is None else deco(obj)

    def get(self, k):
        return self._d[k]


class Instances:
    """Attribute bag standing in for detectron2.structures.Instances (fields + .to())."""

    def __init__(self, image_size, **kw):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kw.items():
            self._fields[k] = v

    def __setattr__(self, k, v):
        self._fields[k] = v

    def __getattr__(self, k):
        f = object.__getattribute__(self, "_fields")
        if k in f:
            return f[k]
        raise AttributeError(k)

    def to(self, device):
        r = Instances(self._image_size)
        for k, v in self._fields.items():
            r._fields[k] = v.to(device) if hasattr(v, "to") else v
        return r

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0


class ImageList:
    """Zero-pad-to-divisibility + image_sizes, standing in for detectron2 ImageList.from_tensors."""

    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        import torch
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        mh = max(s[0] for s in sizes)
        mw = max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            mh = (mh + d - 1) // d * d
            mw = (mw + d - 1) // d * d
        out = tensors[0].new_full((len(tensors), tensors[0].shape[0], mh, mw), pad_value)
        for i, t in enumerate(tensors):
            out[i, :, : t.shape[-2], : t.shape[-1]].copy_(t)
        return ImageList(out, sizes)


_installed = False
BACKBONE_BUILDER = {"fn": None}   # make_golden sets this to a callable(cfg) -> nn.Module


def install():
    """Register stand-in modules and empty package shells; idempotent."""
    global _installed
    if _installed:
        return
    _installed = True
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    # --- torchvision stand-in (mdqe/util/misc.py:21-24,317,475) --------------------------------
    tv = _mod("torchvision", __version__="0.15.0", _is_tracing=lambda: False)
    tv_ops = _mod("torchvision.ops")
    tv_misc = _mod("torchvision.ops.misc", interpolate=F.interpolate)
    tv_ops.misc = tv_misc
    tv.ops = tv_ops
    tv_models = _mod("torchvision.models")
    tv_utils = _mod("torchvision.models._utils", IntermediateLayerGetter=object)
    tv_models._utils = tv_utils
    tv.models = tv_models

    # --- detectron2 stand-ins ------------------------------------------------------------------
    d2 = _mod("detectron2")
    META = _Registry()
    BB = _Registry()

    def build_backbone(cfg):
        assert BACKBONE_BUILDER["fn"] is not None, "set refshim.BACKBONE_BUILDER['fn']"
        return BACKBONE_BUILDER["fn"](cfg)

    _mod("detectron2.modeling", META_ARCH_REGISTRY=META, build_backbone=build_backbone)

    class Boxes:
        """detectron2.structures.Boxes stand-in: a [N,4] xyxy tensor holder."""

        def __init__(self, tensor):
            self.tensor = tensor

        def to(self, device):
            return Boxes(self.tensor.to(device))

        def __len__(self):
            return self.tensor.shape[0]

    class BitMasks:
        """detectron2.structures.BitMasks stand-in with the PUBLIC get_bounding_boxes algorithm (detectron2 v0.6,
        structures/masks.py): tight box [x_min, y_min, x_max + 1, y_max + 1] of the non-zero pixels, zeros for empty masks."""

        def __init__(self, tensor):
            self.tensor = tensor.to(torch.bool)

        def get_bounding_boxes(self):
            boxes = torch.zeros(self.tensor.shape[0], 4, dtype=torch.float32)
            x_any = torch.any(self.tensor, dim=1)
            y_any = torch.any(self.tensor, dim=2)
            for idx in range(self.tensor.shape[0]):
                x = torch.where(x_any[idx, :])[0]
                y = torch.where(y_any[idx, :])[0]
                if len(x) > 0 and len(y) > 0:
                    boxes[idx, :] = torch.as_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1], dtype=torch.float32)
            return Boxes(boxes)

    _mod("detectron2.structures", Instances=Instances, ImageList=ImageList, Boxes=Boxes, BitMasks=BitMasks)
    _mod("detectron2.utils")
    _mod("detectron2.utils.memory", retry_if_cuda_oom=lambda f: f)
    _mod("detectron2.modeling.backbone", Backbone=nn.Module)
    _mod("detectron2.modeling.backbone.build", BACKBONE_REGISTRY=BB)
    _mod("detectron2.modeling.backbone.fpn", FPN=object, LastLevelMaxPool=object)

    class ShapeSpec:
        def __init__(self, channels=None, height=None, width=None, stride=None):
            self.channels, self.height, self.width, self.stride = channels, height, width, stride

    _mod("detectron2.layers", ShapeSpec=ShapeSpec)

    # --- timm stand-in (mdqe/backbone/swin_transformer_v2.py:12-18) ----------------------------
    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple,
         trunc_normal_=lambda t, std=0.02, **k: nn.init.trunc_normal_(t, std=std))

    # --- empty package shells so submodules import without the packages' __init__ --------------
    base = os.path.join(REF_ROOT, "mdqe")
    _pkg("mdqe", base)
    for sub in ("models", "util", "tracking", "backbone"):
        _pkg("mdqe." + sub, os.path.join(base, sub))
    _pkg("mdqe.models.ops", os.path.join(base, "models", "ops"))
    _pkg("mdqe.models.ops.functions", os.path.join(base, "models", "ops", "functions"))
    _pkg("mdqe.models.ops.modules", os.path.join(base, "models", "ops", "modules"))

    # --- stand-in for the CUDA-only extension ---------------------------------------------------
    msda = _mod("MultiScaleDeformableAttention")

    def ms_deform_attn_forward(value, shapes, level_start, loc, attn, im2col_step):
        func = importlib.import_module("mdqe.models.ops.functions.ms_deform_attn_func")
        return func.ms_deform_attn_core_pytorch(value, shapes.tolist(), loc, attn)

    msda.ms_deform_attn_forward = ms_deform_attn_forward

    func = importlib.import_module("mdqe.models.ops.functions.ms_deform_attn_func")
    sys.modules["mdqe.models.ops.functions"].MSDeformAttnFunction = func.MSDeformAttnFunction
    sys.modules["mdqe.models.ops.functions"].ms_deform_attn_core_pytorch = func.ms_deform_attn_core_pytorch
    modm = importlib.import_module("mdqe.models.ops.modules.ms_deform_attn")
    sys.modules["mdqe.models.ops.modules"].MSDeformAttn = modm.MSDeformAttn

    # models package surface used by mdqe/mdqe.py:14 (criterion/matcher are training-only and
    # pull detectron2 point_rend; give inert placeholders)
    mm = sys.modules["mdqe.models"]
    mm.mdqe = importlib.import_module("mdqe.models.mdqe").mdqe
    mm.Transformer_Enc = importlib.import_module("mdqe.models.transformer_enc").Transformer_Enc
    mm.Transformer_Dec = importlib.import_module("mdqe.models.transformer_dec").Transformer_Dec

    class _Inert(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    mm.SetCriterion = _Inert
    mm.HungarianMatcher = _Inert
    mm.ClipPeakMatcher = _Inert
    trk = importlib.import_module("mdqe.tracking.OverTracker")
    sys.modules["mdqe.tracking"].Clips = trk.Clips
    sys.modules["mdqe.tracking"].OverTracker = trk.OverTracker


def ref(name):
    """Import a reference submodule, e.g. ref('mdqe.models.transformer_enc')."""
    install()
    return importlib.import_module(name)
