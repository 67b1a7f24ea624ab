"""Deterministic synthetic weights keyed by parameter name  (TEST INFRASTRUCTURE).

Golden fixtures would be tens of MB if they stored the weights the reference modules were run with
(the reference's mask head needs hidden_dim in {192, 256, ...}: GroupNorm(24|32),
mdqe/models/segmentation.py:104).  Instead every float tensor of a module is overwritten, before
the reference is executed, by values drawn from numpy's *frozen* legacy generator
(`np.random.RandomState`, bit-stream guaranteed stable across numpy versions) seeded by
crc32(name) ^ seed.  A fixture then only stores the manifest (names + shapes) plus inputs/outputs;
tests re-create the identical state dict with `synth_state`.

This also removes the zero-init trap: attention_weights / sampling_offsets /
sampling_grid_offsets get non-zero values and the classifier biases are not -4.6.
"""
import zlib
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

# buffers that are *defined* by formulas in the reference and must be kept, not synthesised
KEEP_SUFFIXES = ("cross_attn.sampling_offsets", "temp_attn_inst.sampling_offsets", "lvl_spatial_scales",
                 "query_relpos_grid", "num_batches_tracked", "relative_coords_table")


def _kind(name: str, shape: Tuple[int, ...]) -> str:
    if name.endswith(KEEP_SUFFIXES):
        return "keep"
    if name.endswith("running_var"):
        return "var"
    if name.endswith(("running_mean",)):
        return "bias"
    if name.endswith(("attention_weights.weight",)):
        return "w_small"
    if name.endswith(("sampling_offsets.weight", "sampling_grid_offsets.weight")):
        return "w_off"
    if name.endswith(("sampling_offsets.bias", "sampling_grid_offsets.bias")):
        return "b_off"
    if name.endswith("level_embed"):
        return "normal"
    if len(shape) == 1 and name.endswith("weight"):
        return "gamma"                     # LayerNorm / GroupNorm / FrozenBN scale
    if len(shape) == 1:
        return "bias"
    return "xavier"


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    k = _kind(name, tuple(shape))
    n = int(np.prod(shape)) if len(shape) else 1
    if k == "xavier":
        fan_out = shape[0]
        fan_in = int(np.prod(shape[1:]))
        a = np.sqrt(6.0 / (fan_in + fan_out))
        v = rs.uniform(-a, a, n)
    elif k == "gamma":
        v = 1.0 + 0.1 * rs.standard_normal(n)
    elif k == "var":
        v = 0.5 + np.abs(rs.standard_normal(n))
    elif k == "bias":
        v = 0.05 * rs.standard_normal(n)
    elif k == "w_small":
        v = 0.08 * rs.standard_normal(n)
    elif k == "w_off":
        v = 0.06 * rs.standard_normal(n)
    elif k == "b_off":
        v = 0.8 * rs.standard_normal(n)
    elif k == "normal":
        v = rs.standard_normal(n)
    else:
        raise ValueError(k)
    return torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape))


def apply_synth(module: torch.nn.Module, seed: int = 0, prefix: str = "") -> List[Tuple[str, Tuple[int, ...]]]:
    """Overwrite every float param/buffer of `module` in place; returns the manifest."""
    manifest = []
    first = {}          # storage -> first name (the reference's checkpoint holds aliased tensors under
    #                     several names, e.g. transformer_dec.bbox_embed == transformer_dec.decoder.bbox_embed)
    with torch.no_grad():
        for name, t in module.state_dict().items():
            full = prefix + name
            if not t.dtype.is_floating_point or _kind(full, tuple(t.shape)) == "keep":
                continue
            key = (t.data_ptr(), tuple(t.shape))
            if key in first:
                manifest.append((full, "=" + first[key]))
                continue
            first[key] = full
            t.copy_(synth_tensor(full, tuple(t.shape), seed))
            manifest.append((full, tuple(t.shape)))
    return manifest


def manifest_to_arrays(manifest) -> Dict[str, np.ndarray]:
    return {"manifest_names": np.array([m[0] for m in manifest]),
            "manifest_shapes": np.array([m[1] if isinstance(m[1], str) else ",".join(str(int(s)) for s in m[1])
                                         for m in manifest])}


def synth_state(names: Iterable[str], shapes: Iterable[str], seed: int = 0) -> Dict[str, torch.Tensor]:
    sd = {}
    for n, s in zip(names, shapes):
        n, s = str(n), str(s)
        if s.startswith("="):
            sd[n] = sd[s[1:]]
            continue
        shp = tuple(int(v) for v in s.split(",")) if s != "" else ()
        sd[n] = synth_tensor(n, shp, seed)
    return sd
