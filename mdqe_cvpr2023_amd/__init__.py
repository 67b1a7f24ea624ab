"""mdqe_cvpr2023_amd -- MI355X-native MDQE eval-only inference path.

Hand-written HIP kernels for gfx950 live in csrc/ behind the C ABI declared in include/mdqe_hip.h;
this package is the host-side mirror of the reference's operator surface for that path
(SURVEY.md §8b).  It requires the compiled library: there is NO CPU or PyTorch fallback.
"""
from ._lib import lib, load_library, LibraryMissing  # noqa: F401

__all__ = ["lib", "load_library", "LibraryMissing"]
