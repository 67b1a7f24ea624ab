"""detectron2-facing surface of the reference package for the eval path (SURVEY.md §8b B-model):

    from mdqe_cvpr2023_amd import add_mdqe_config, add_swinl_config, MDQE

mirrors `from mdqe import add_mdqe_config, add_swinl_config, MDQE` (mdqe/__init__.py:1-4) so the reference's
`train_net.py --eval-only` / `demo/demo.py` set-up code (`setup(args)`, train_net.py:224-235) works unchanged.  The
functions only declare config keys with the reference's defaults (mdqe/config.py:5-85, mdqe/backbone/config.py:60-75);
they need detectron2's yacs `CfgNode` and raise ImportError when detectron2 is not installed (it is not in this image).
"""


def _CN():
    try:
        from detectron2.config import CfgNode
    except Exception as e:                       # pragma: no cover - detectron2 absent in the build image
        raise ImportError("add_*_config needs detectron2 (yacs CfgNode); use mdqe_cvpr2023_amd.config.MDQEConfig instead") from e
    return CfgNode


MDQE_DEFAULTS = dict(
    NUM_CLASSES=80, BOX_WEIGHT=2.0, MASK_WEIGHT=4.0, DICE_WEIGHT=4.0, DEEP_SUPERVISION=True, NO_OBJECT_WEIGHT=1, MASK_STRIDE=4,
    MATCH_STRIDE=4, MASK_DIM=32, NUM_MASK_LAYERS=1, NHEADS=8, DROPOUT=0.1, MLP_RATIO=4, ENC_LAYERS=6, DEC_LAYERS=6, PRE_NORM=False,
    HIDDEN_DIM=256, NUM_OBJECT_QUERIES=200, NUM_FEATURE_LEVELS=4, ENC_NUM_POINTS=4, DEC_NUM_POINTS=4, DEC_TEMPORAL=True,
    QUERY_EMBED_DIM=64, WINDOW_INTER_FRAME_ASSOCIATION=5, INTERINST_MASK_LOSS_ENABLED=True, INTERINST_MASK_THRESHOLD=0.1,
    CLIP_STRIDE=1, SAMPLING_FRAME_NUM_TEST=5, WINDOW_FRAME_NUM_TEST=20, MAX_NUM_INSTANCES=50, MERGE_ON_CPU=False, MULTI_CLS_ON=True,
    APPLY_CLS_THRES=0.05)


def add_mdqe_config(cfg):
    """Same keys/defaults as mdqe/config.py:5-85."""
    CN = _CN()
    cfg.DATASETS.DATASET_RATIO = []
    cfg.MODEL.MDQE = CN()
    for k, v in MDQE_DEFAULTS.items():
        setattr(cfg.MODEL.MDQE, k, v)
    cfg.INPUT.PRETRAIN_FRAME_NUM = 1
    cfg.INPUT.SAMPLING_FRAME_NUM = 3
    cfg.INPUT.SAMPLING_FRAME_RANGE = 10
    cfg.INPUT.SAMPLING_FRAME_SHUFFLE = False
    cfg.INPUT.AUGMENTATIONS = []
    cfg.INPUT.PSEUDO = CN()
    cfg.INPUT.PSEUDO.AUGMENTATIONS = ["rotation"]
    cfg.INPUT.PSEUDO.MIN_SIZE_TRAIN = (480, 512, 544, 576, 608, 640, 672, 704, 736, 768)
    cfg.INPUT.PSEUDO.MAX_SIZE_TRAIN = 768
    cfg.INPUT.PSEUDO.MIN_SIZE_TRAIN_SAMPLING = "choice_by_clip"
    cfg.INPUT.PSEUDO.CROP = CN()
    cfg.INPUT.PSEUDO.CROP.ENABLED = False
    cfg.INPUT.PSEUDO.CROP.TYPE = "absolute_range"
    cfg.INPUT.PSEUDO.CROP.SIZE = (384, 600)
    cfg.INPUT.LSJ_AUG = CN()
    cfg.INPUT.LSJ_AUG.ENABLED = False
    cfg.INPUT.LSJ_AUG.IMAGE_SIZE = 1024
    cfg.INPUT.LSJ_AUG.MIN_SCALE = 0.1
    cfg.INPUT.LSJ_AUG.MAX_SCALE = 2.0
    cfg.SOLVER.OPTIMIZER = "ADAMW"
    cfg.SOLVER.BACKBONE_MULTIPLIER = 0.1
    cfg.SOLVER.NUM_PRETRAIN_FRAMES = 1


def _add_swin(cfg, embed, depths, heads, window, drop_path):
    CN = _CN()
    cfg.MODEL.SWIN = CN()
    cfg.MODEL.SWIN.EMBED_DIM = embed
    cfg.MODEL.SWIN.OUT_FEATURES = ["stage3", "stage4", "stage5"]
    cfg.MODEL.SWIN.DEPTHS = depths
    cfg.MODEL.SWIN.NUM_HEADS = heads
    cfg.MODEL.SWIN.WINDOW_SIZE = window
    cfg.MODEL.SWIN.MLP_RATIO = 4
    cfg.MODEL.SWIN.DROP_PATH_RATE = drop_path
    cfg.MODEL.SWIN.APE = False
    cfg.MODEL.BACKBONE.FREEZE_AT = -1
    cfg.MODEL.FPN.TOP_LEVELS = 2
    cfg.SOLVER.OPTIMIZER = "AdamW"


def add_swinl_config(cfg):
    """mdqe/backbone/config.py:60-75."""
    _add_swin(cfg, 192, [2, 2, 18, 2], [6, 12, 24, 48], 24, 0.2)


def add_swinb_config(cfg):
    _add_swin(cfg, 128, [2, 2, 18, 2], [4, 8, 16, 32], 7, 0.2)


def add_swins_config(cfg):
    _add_swin(cfg, 96, [2, 2, 18, 2], [3, 6, 12, 24], 7, 0.2)


def add_swint_config(cfg):
    _add_swin(cfg, 96, [2, 2, 6, 2], [3, 6, 12, 24], 7, 0.2)


# ---- result containers of the COCO single-image branch -----------------------------------------------------------------
try:                                              # the real ones when detectron2 is installed
    from detectron2.structures import Boxes, Instances
except Exception:                                 # detectron2 absent in this image: attribute bags with the fields the evaluator reads
    class Boxes:
        def __init__(self, tensor):
            self.tensor = tensor

        def to(self, device):
            return Boxes(self.tensor.to(device))

        def __len__(self):
            return self.tensor.shape[0]

    class Instances:
        def __init__(self, image_size, **fields):
            object.__setattr__(self, "_image_size", tuple(image_size))
            object.__setattr__(self, "_fields", dict(fields))

        @property
        def image_size(self):
            return self._image_size

        def __setattr__(self, k, v):
            self._fields[k] = v

        def __getattr__(self, k):
            f = object.__getattribute__(self, "_fields")
            if k in f:
                return f[k]
            raise AttributeError(k)

        def has(self, k):
            return k in self._fields

        def get_fields(self):
            return self._fields

        def to(self, device):
            r = Instances(self._image_size)
            for k, v in self._fields.items():
                r._fields[k] = v.to(device) if hasattr(v, "to") else v
            return r

        def __len__(self):
            for v in self._fields.values():
                return len(v)
            return 0
