"""from_d2_cfg reads the reference's yacs keys (mdqe/mdqe.py:64-103,176-192; mdqe/config.py) -- checked with an attribute
tree carrying the values of configs/R50_ovis_360.yaml / swinl_ovis.yaml (yacs / detectron2 are not installed here)."""
from types import SimpleNamespace as NS

from mdqe_cvpr2023_amd.config import PRESETS, from_d2_cfg


def _cfg(**over):
    m = NS(NUM_CLASSES=25, MASK_STRIDE=4, MATCH_STRIDE=4, HIDDEN_DIM=256, NUM_OBJECT_QUERIES=200, WINDOW_INTER_FRAME_ASSOCIATION=5,
           QUERY_EMBED_DIM=64, NHEADS=8, ENC_LAYERS=6, DEC_LAYERS=6, NUM_FEATURE_LEVELS=4, DEC_NUM_POINTS=4, ENC_NUM_POINTS=4,
           DEC_TEMPORAL=True, MLP_RATIO=4, CLIP_STRIDE=1, MERGE_ON_CPU=True, MULTI_CLS_ON=True, APPLY_CLS_THRES=0.1,
           SAMPLING_FRAME_NUM_TEST=4, WINDOW_FRAME_NUM_TEST=30, MAX_NUM_INSTANCES=120)
    cfg = NS(INPUT=NS(SAMPLING_FRAME_NUM=4), DATASETS=NS(TEST=("ytvis_ovis_val",)), TEST=NS(DETECTIONS_PER_IMAGE=15),
             MODEL=NS(DEVICE="cuda", PIXEL_MEAN=[123.675, 116.280, 103.530], PIXEL_STD=[58.395, 57.120, 57.375], MDQE=m,
                      RESNETS=NS(DEPTH=50), BACKBONE=NS(NAME="build_resnet_backbone")))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def test_r50_ovis_360_keys():
    c = from_d2_cfg(_cfg())
    p = PRESETS["R50_ovis_360"]
    for f in ("backbone", "hidden_dim", "nheads", "enc_layers", "dec_layers", "n_levels", "n_frames", "num_classes", "num_queries",
              "n_frames_test", "n_frames_window_test", "n_max_inst", "apply_cls_thres", "match_stride", "clip_stride"):
        assert getattr(c, f) == getattr(p, f), f
    assert c.n_query == 196 and c.is_coco is False and c.multi_cls is True and c.device == "cuda"


def test_coco_test_set_switches_the_image_branch():
    c = from_d2_cfg(_cfg(DATASETS=NS(TEST=("coco_2017_val",))))
    assert c.is_coco is True


def test_swin_backbone_keys():
    cfg = _cfg()
    cfg.MODEL.BACKBONE = NS(NAME="build_swinv2_backbone")
    cfg.MODEL.SWIN = NS(EMBED_DIM=192, DEPTHS=[2, 2, 18, 2], NUM_HEADS=[6, 12, 24, 48], WINDOW_SIZE=12, MLP_RATIO=4.0)
    c = from_d2_cfg(cfg)
    assert c.backbone == "SwinV2" and c.swin_depths == (2, 2, 18, 2) and c.backbone_channels == (384, 768, 1536)
