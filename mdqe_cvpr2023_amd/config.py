"""Hyper-parameters of the eval path and the shipped presets.

Mirrors the keys the reference reads in MDQE.__init__ (mdqe/mdqe.py:63-192) from its yacs config
(mdqe/config.py:5-85, configs/R50_coco.yaml, R50_ovis_360.yaml, R50_ovis_720.yaml, swinl_ovis.yaml).
`from_d2_cfg` accepts the reference's own CfgNode (or any attribute tree with the same keys), so
`MDQE(cfg)` works unchanged when detectron2 is installed.
"""
import math
from dataclasses import dataclass, replace
from typing import Tuple


@dataclass
class MDQEConfig:
    backbone: str = "R50"                     # "R50" | "R101" | "SwinV2" | "custom"
    swin_embed_dim: int = 192                 # MODEL.SWIN.* (mdqe/backbone/config.py:60-75, configs/swinl_coco.yaml)
    swin_depths: Tuple[int, ...] = (2, 2, 18, 2)
    swin_heads: Tuple[int, ...] = (6, 12, 24, 48)
    swin_window: int = 12
    swin_mlp_ratio: float = 4.0
    backbone_channels: Tuple[int, ...] = (512, 1024, 2048)
    backbone_strides: Tuple[int, ...] = (8, 16, 32)
    hidden_dim: int = 256
    nheads: int = 8
    enc_layers: int = 6
    dec_layers: int = 6
    n_levels: int = 4
    enc_points: int = 4
    dec_points: int = 4
    n_frames: int = 4                         # INPUT.SAMPLING_FRAME_NUM
    num_classes: int = 25
    num_queries: int = 200
    query_embed_dim: int = 64
    window_inter_frame_asso: float = 5
    mlp_ratio: float = 4
    dec_temporal: bool = True
    clip_stride: int = 1
    n_frames_test: int = 4
    n_frames_window_test: int = 30
    n_max_inst: int = 120
    apply_cls_thres: float = 0.1
    detections_per_image: int = 15
    match_stride: int = 4
    merge_on_cpu: bool = False
    size_divisibility: int = 32
    pixel_mean: Tuple[float, ...] = (123.675, 116.280, 103.530)
    pixel_std: Tuple[float, ...] = (58.395, 57.120, 57.375)
    min_size_test: int = 360                  # INPUT.MIN_SIZE_TEST
    max_size_test: int = 1333                 # INPUT.MAX_SIZE_TEST (detectron2 default; the configs do not set it)
    is_coco: bool = False                     # DATASETS.TEST[0].startswith("coco") (mdqe/mdqe.py:70): single-image branch
    multi_cls: bool = True                    # MODEL.MDQE.MULTI_CLS_ON (mdqe/mdqe.py:187)
    device: str = "cuda"

    @property
    def n_query(self):
        return int(math.sqrt(self.num_queries)) ** 2     # 200 -> 196, mdqe/mdqe.py:77-78

    @property
    def n_bins(self):
        return int(math.sqrt(self.n_query))

    @property
    def mask_dim(self):
        return self.hidden_dim // 8

    @property
    def d_ffn(self):
        return int(self.hidden_dim * self.mlp_ratio)


R50_OVIS_360 = MDQEConfig()
R50_OVIS_720 = replace(R50_OVIS_360, n_frames_window_test=20, merge_on_cpu=True, apply_cls_thres=0.2, min_size_test=640)
SWINL_OVIS = MDQEConfig(backbone="SwinV2", backbone_channels=(384, 768, 1536), hidden_dim=192, n_frames=2, n_frames_test=2,
                        n_frames_window_test=20, merge_on_cpu=True, apply_cls_thres=0.1, min_size_test=480)
PRESETS = {"R50_ovis_360": R50_OVIS_360, "R50_ovis_720": R50_OVIS_720, "swinl_ovis": SWINL_OVIS}


def from_d2_cfg(cfg) -> MDQEConfig:
    m = cfg.MODEL.MDQE
    depth = getattr(getattr(cfg.MODEL, "RESNETS", None), "DEPTH", 50)
    name = getattr(getattr(cfg.MODEL, "BACKBONE", None), "NAME", "build_resnet_backbone")
    swin = {}
    if "swinv2" in name:
        sw = cfg.MODEL.SWIN
        swin = dict(backbone="SwinV2", swin_embed_dim=sw.EMBED_DIM, swin_depths=tuple(sw.DEPTHS), swin_heads=tuple(sw.NUM_HEADS),
                    swin_window=sw.WINDOW_SIZE, swin_mlp_ratio=sw.MLP_RATIO,
                    backbone_channels=tuple(sw.EMBED_DIM * 2 ** i for i in (1, 2, 3)))
    return MDQEConfig(
        **({"backbone": "R%d" % depth} if not swin else swin),
        hidden_dim=m.HIDDEN_DIM, nheads=m.NHEADS, enc_layers=m.ENC_LAYERS, dec_layers=m.DEC_LAYERS,
        n_levels=m.NUM_FEATURE_LEVELS, enc_points=m.ENC_NUM_POINTS, dec_points=m.DEC_NUM_POINTS,
        n_frames=cfg.INPUT.SAMPLING_FRAME_NUM, num_classes=m.NUM_CLASSES, num_queries=m.NUM_OBJECT_QUERIES,
        query_embed_dim=m.QUERY_EMBED_DIM, window_inter_frame_asso=m.WINDOW_INTER_FRAME_ASSOCIATION,
        mlp_ratio=m.MLP_RATIO, dec_temporal=m.DEC_TEMPORAL, clip_stride=m.CLIP_STRIDE,
        n_frames_test=m.SAMPLING_FRAME_NUM_TEST, n_frames_window_test=m.WINDOW_FRAME_NUM_TEST,
        n_max_inst=m.MAX_NUM_INSTANCES, apply_cls_thres=m.APPLY_CLS_THRES,
        detections_per_image=cfg.TEST.DETECTIONS_PER_IMAGE, match_stride=m.MATCH_STRIDE,
        merge_on_cpu=m.MERGE_ON_CPU, pixel_mean=tuple(cfg.MODEL.PIXEL_MEAN), pixel_std=tuple(cfg.MODEL.PIXEL_STD),
        is_coco=str(cfg.DATASETS.TEST[0]).startswith("coco") if len(getattr(cfg.DATASETS, "TEST", ())) else False,
        multi_cls=bool(getattr(m, "MULTI_CLS_ON", True)), min_size_test=int(getattr(cfg.INPUT, "MIN_SIZE_TEST", 360)),
        max_size_test=int(getattr(cfg.INPUT, "MAX_SIZE_TEST", 1333)), device=str(cfg.MODEL.DEVICE))
