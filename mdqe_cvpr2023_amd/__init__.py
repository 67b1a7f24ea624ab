"""mdqe_cvpr2023_amd -- MI355X-native MDQE eval-only inference path.

Hand-written HIP kernels for gfx950 live in csrc/ behind the C ABI declared in include/mdqe_hip.h;
this package is the host-side mirror of the reference's operator surface for that path
(SURVEY.md §8b).  It requires the compiled library: there is NO CPU or PyTorch fallback.
"""
from ._lib import lib, load_library, LibraryMissing  # noqa: F401
from .d2_compat import add_mdqe_config, add_swinl_config, add_swinb_config, add_swins_config, add_swint_config  # noqa: F401


def __getattr__(name):                 # `from mdqe_cvpr2023_amd import MDQE` without importing torch.nn at package import
    if name in ("MDQE", "MDQE_MI355X", "register_with_detectron2"):
        from . import meta_arch
        return getattr(meta_arch, name)
    if name in ("build_swinv2_backbone", "SwinTransformerV2"):
        from . import backbone
        return getattr(backbone, name)
    raise AttributeError(name)


__all__ = ["lib", "load_library", "LibraryMissing", "MDQE", "MDQE_MI355X", "register_with_detectron2", "add_mdqe_config", "add_swinl_config", "add_swinb_config",
           "add_swins_config", "add_swint_config", "build_swinv2_backbone", "SwinTransformerV2"]
