"""ctypes binding of libmdqe_hip.so (C ABI: include/mdqe_hip.h).  Fails loudly when absent."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDQE_HIP_LIB") or os.path.join(_HERE, "csrc", "libmdqe_hip.so")   # env override: A/B kernel builds (tools/)

from ctypes import c_long
i, f, p, l = c_int, c_float, c_void_p, c_long

# name -> argtypes; every function returns int status (include/mdqe_hip.h)
SIGNATURES = {
    "mdqe_msda_forward_f32": [p, p, p, p, p, i, i, i, i, i, i, i, p, p],
    "mdqe_msda_backward_f32": [p, p, p, p, p, p, i, i, i, i, i, i, i, p, p, p, p],
    "mdqe_msda_forward_f64": [p, p, p, p, p, i, i, i, i, i, i, i, p, p],
    "mdqe_msda_backward_f64": [p, p, p, p, p, p, i, i, i, i, i, i, i, p, p, p, p],
    "mdqe_msda_forward_grouped_f32": [p, p, p, p, p, i, i, i, i, i, i, i, i, f, p, p],
    "mdqe_msda_fused_f32": [p, l, l, p, p, l, p, l, p, l, i, i, p, p, p, p, i, i, i, i, i, i, i, f, p, l, l, p],
    "mdqe_trk_siou_f32": [p, l, i, p, l, i, l, p, p],
    "mdqe_trk_accumulate_f32": [p, l, p, l, p, l, l, i, p, p, i, p],
    "mdqe_trk_window_mean_f32": [p, p, l, i, i, i, l, p, p],
    "mdqe_trk_carry_f32": [p, p, l, i, i, i, l, p, p],
    "mdqe_lsap_f64": [p, i, i, i, p, p, p],
    "mdqe_tracker_create": [i, i, i, i, i, i, f, p],
    "mdqe_tracker_destroy": [p],
    "mdqe_tracker_state": [p, p, p, p],
    "mdqe_tracker_overlap": [p, i, i, p, p, p, p],
    "mdqe_tracker_decide": [p, i, i, i, p, p, p, p, p, p, p, p, p, p],
    "mdqe_tracker_result": [p, i, p, p, p, p],
    "mdqe_tracker_update": [p, p, p, l, i, i, i, p, p, p, p, l, p, p, p],
    "mdqe_tracker_update_many": [p, p, p, l, i, p, p, p, p, p, p, p, p, p, p, p, p],
    "mdqe_tracker_get_result": [p, i, p, p, l, p, p, p, p, p, p],
    "mdqe_clip_assoc_f32": [p, i, i, p, i, i, i, f, i, p, p],
    "mdqe_clip_gather_init_f32": [p, p, p, p, i, i, i, i, i, p, p, p, p],
    "mdqe_box_refine_f32": [p, p, i, i, i, i, i, p, p, p],
    "mdqe_add_rows_f32": [p, l, p, l, p, l, l, i, p],
    "mdqe_time_fuse_f32": [p, p, i, i, i, i, p, p, p, p],
    "mdqe_box_head_refine_f32": [p, l, p, p, p, i, i, i, i, i, i, p, p, p],
    "mdqe_time_fuse_dot_f32": [p, p, p, p, i, i, i, i, p, p],
    "mdqe_clip_select_f32": [p, p, i, i, i, i, f, i, p, p, p, p, p, p, p],
    "mdqe_dyn_mask_nms_f32": [p, p, p, i, i, i, i, i, i, p, p, p, p, p, p, p, p, p, p, p],
    "mdqe_clip_finalize_f32": [p, p, p, p, p, i, i, i, i, f, p, p, p, p, p, p],
    "mdqe_rows_gather_f32": [p, p, i, l, p, p],
    "mdqe_mha_small_f32": [p, l, p, l, p, l, i, i, i, i, p],
    "mdqe_query_select_f32": [p, i, i, i, i, i, p, p, p],
    "mdqe_sample_levels_mean_f32": [p, i, l, i, p, i, p, p, p, i, p, p],
    "mdqe_image_mask_stats_f32": [p, i, i, i, i, i, i, p, p],
    "mdqe_image_final_masks_u8": [p, i, p, i, i, i, i, i, i, i, p, p],
    "mdqe_final_masks_rle": [p, i, p, i, i, i, i, i, i, i, i, i, p, p, p],
    "mdqe_final_masks_u8": [p, i, p, i, i, i, i, i, i, i, i, p, l, i, p],
    "mdqe_set_gemm_precision": [i],
    "mdqe_set_gemm_precision_thread": [i],
    "mdqe_layernorm_post_f32": [p, p, p, p, p, l, i, f, p],
    "mdqe_gemm_nt_swin_f32": [p, l, p, p, p, l, i, i, i, i, i, i, i, p],
    "mdqe_layernorm_swin_scatter_f32": [p, p, p, p, p, i, i, i, i, i, i, f, p],
    "mdqe_patch4_im2col_f32": [p, i, l, i, i, i, i, i, p, p, p, p],
    "mdqe_swin_window_f32": [p, p, p, i, i, i, i, i, i, i, p],
    "mdqe_window_attn_f32": [p, l, p, l, i, i, i, i, p, p, p, i, p],
    "mdqe_patch_merge_gather_f32": [p, p, i, i, i, i, p],
    "mdqe_gemm_nt_f32": [p, l, p, p, p, l, i, i, i, i, i, p, l, i, i, p, i, i, i, p, p, p],
    "mdqe_f16x3_split_f32": [p, l, p, p],
    "mdqe_debug_gemm_stamps": [p],
    "mdqe_debug_gemm_variant": [i],
    "mdqe_debug_gemm_tile_rule": [i],
    "mdqe_debug_gemm_rows_dot": [i],
    "mdqe_debug_gemm_stagger": [i],
    "mdqe_debug_gemm_fast_epilogue": [i],
    "mdqe_debug_gemm_stages": [i],
    "mdqe_debug_gemm_lds_pad": [i],
    "mdqe_debug_trk_siou_blocks": [i],
    "mdqe_trk_siou_host_f32": [p, l, i, p, l, i, l, p, p, p, i, p],
    "mdqe_trk_wait_counts": [p, i, i, i, p, p],
    "mdqe_debug_trk_times": [p, i],
    "mdqe_debug_trk_fast": [i],
    "mdqe_debug_trk_spin_us": [i],
    "mdqe_debug_msda_dec_stage_kb": [i],
    "mdqe_debug_window_attn_variant": [i],
    "mdqe_debug_mha_variant": [i],
    "mdqe_debug_msda_xcd_order": [i],
    "mdqe_debug_msda_dec_staged": [i],
    "mdqe_debug_msda_patch": [i],
    "mdqe_debug_msda_dec_wpe8": [i],
    "mdqe_debug_msda_op_staged": [i],
    "mdqe_debug_msda_tp_staged": [i],
    "mdqe_debug_msda_stage_kb": [i],
    "mdqe_debug_msda_variant": [i],
    "mdqe_mask_row_stats_f32": [p, i, i, i, i, i, p, p, p, p],
    "mdqe_conv2d_nhwc_f32": [p, l, p, p, p, l, i, i, i, i, i, i, i, i, i, i, p, l, i, i, p, i, p, p],
    "mdqe_layernorm_f32": [p, p, p, p, p, l, i, f, p],
    "mdqe_gemm_ln_f32": [p, l, p, p, p, l, i, i, i, p, l, p, p, f, p],
    "mdqe_gemm_ln2_f32": [p, l, p, p, p, l, i, i, i, p, l, p, p, p, p, p, l, f, p],
    "mdqe_gemm_nt_side_f32": [p, l, p, p, p, l, i, i, i, p, p, i, p, p],
    "mdqe_gemm_nt_cat2_f32": [p, l, i, p, l, i, i, i, i, i, i, i, p, p, p, l, i, i, p],
    "mdqe_groupnorm_nhwc_f32": [p, l, l, p, l, l, i, i, i, i, p, p, f, i, p, p],
    "mdqe_resize_pil_bilinear_u8": [p, l, i, i, i, i, i, i, p, p, p, i, p, p, p, i, p, p],
    "mdqe_stem_im2col_f32": [p, i, l, i, i, i, i, i, p, p, p, p],
    "mdqe_stem_conv_f32": [p, i, l, i, i, i, i, i, p, p, p, p, p, p],
    "mdqe_maxpool3x3s2_nhwc_f32": [p, p, i, i, i, i, p],
    "mdqe_upsample_nearest_add_nhwc_f32": [p, p, p, i, i, i, i, i, i, p],
    "mdqe_dwconv5x5_nhwc_f32": [p, p, p, p, i, i, i, i, i, p, p, p],
    "mdqe_dwconv5x5_c256_f32": [p, p, p, p, i, i, i, p],
    "mdqe_dwconv5x5_up2_c256_f32": [p, p, p, p, p, p, i, i, i, p],
}


ABI_VERSION = 6                     # include/mdqe_hip.h MDQE_ABI_VERSION this binding was written against


class LibraryMissing(RuntimeError):
    pass


class MdqeError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise LibraryMissing(
            f"{path} not found: build it with `make -C mdqe_cvpr2023_amd/csrc` (or "
            f"`python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU/PyTorch fallback.")
    # torch must bring ITS HIP runtime (torch/lib/libamdhip64.so) into the process first: the streams and device
    # pointers we are handed belong to that runtime, and a libmdqe_hip.so loaded earlier would bind to /opt/rocm's copy.
    import torch  # noqa: F401
    h = ctypes.CDLL(path)
    h.mdqe_version.restype = c_int
    h.mdqe_abi_version.restype = c_int                 # AttributeError on a library older than the versioned ABI: loud by design
    h.mdqe_abi_version.argtypes = []
    if h.mdqe_abi_version() != ABI_VERSION:
        raise MdqeError(f"{path} implements ABI {h.mdqe_abi_version()}, this binding was written against ABI {ABI_VERSION} "
                        f"(include/mdqe_hip.h MDQE_ABI_VERSION): rebuild with `make -C mdqe_cvpr2023_amd/csrc`")
    h.mdqe_strerror.restype = c_char_p
    h.mdqe_strerror.argtypes = [c_int]
    h.mdqe_get_gemm_precision.restype = c_int
    h.mdqe_get_gemm_precision.argtypes = []
    h.mdqe_nms_workspace_floats.restype = c_long
    h.mdqe_nms_workspace_floats.argtypes = [c_int]
    h.mdqe_dyn_mask_workspace_floats.restype = c_long
    h.mdqe_dyn_mask_workspace_floats.argtypes = [c_int, c_int, c_int, c_int]
    h.mdqe_groupnorm_workspace_bytes.restype = c_long
    h.mdqe_groupnorm_workspace_bytes.argtypes = [c_int, c_int]
    for name, args in SIGNATURES.items():
        fn = getattr(h, name)          # AttributeError if the symbol is missing: loud by design
        fn.argtypes = args
        fn.restype = c_int
    if path == LIB_PATH:
        _lib = h
    return h


class _Lazy:
    def __getattr__(self, name):
        return getattr(load_library(), name)


lib = _Lazy()


def check(code, what=""):
    if code != 0:
        raise MdqeError(f"{what}: {load_library().mdqe_strerror(code).decode()} (code {code})")


def ptr(t):
    """Device pointer of a torch tensor as a plain int (None -> NULL); the argtypes convert it, no ctypes object per argument."""
    return None if t is None else t.data_ptr()


def raw_stream(device=None):
    """hipStream_t (int) of torch's current stream.  Every kernel launch asks for it, so it takes the raw accessor
    (~0.3 us) instead of building a torch.cuda.Stream object (~8 us)."""
    import torch
    if device is None:
        idx = torch._C._cuda_getDevice()
    else:
        idx = device if isinstance(device, int) else torch.device(device).index
        if idx is None:
            idx = torch._C._cuda_getDevice()
    return torch._C._cuda_getCurrentRawStream(idx)


def cur_stream(device=None):
    return raw_stream(device)
