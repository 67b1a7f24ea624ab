"""CPU: the detectron2-facing surface (SURVEY §8b B-model i) EXECUTED -- detectron2 is not installed in this image, so a subprocess
provides minimal stand-ins for `detectron2.config.CfgNode` (attribute tree) and `detectron2.modeling.META_ARCH_REGISTRY`, then runs
what the reference's `setup(args)` runs (train_net.py:224-235): `add_mdqe_config(cfg)`, `add_swinl_config(cfg)`, the YAML overrides of
configs/R50_ovis_360.yaml / swinl_ovis.yaml applied by hand, `MDQE(cfg)` through the registry."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import sys, types
    class CN(dict):                                   # yacs-like attribute tree
        def __getattr__(self, k):
            try: return self[k]
            except KeyError: raise AttributeError(k)
        def __setattr__(self, k, v): self[k] = v
    class Registry(dict):
        def register(self, obj=None):
            self[obj.__name__] = obj
            return obj
        def get(self, name): return self[name]
    d2 = types.ModuleType("detectron2"); d2.__path__ = []
    cfgm = types.ModuleType("detectron2.config"); cfgm.CfgNode = CN
    mod = types.ModuleType("detectron2.modeling"); mod.META_ARCH_REGISTRY = Registry()
    sys.modules.update({"detectron2": d2, "detectron2.config": cfgm, "detectron2.modeling": mod})

    from mdqe_cvpr2023_amd import add_mdqe_config, add_swinl_config, MDQE
    from mdqe_cvpr2023_amd.config import PRESETS
    assert mod.META_ARCH_REGISTRY.get("MDQE") is MDQE      # @META_ARCH_REGISTRY.register() class MDQE (mdqe/mdqe.py:60-63)

    def base_cfg():                                   # the detectron2 defaults the MDQE keys hang off (get_cfg())
        c = CN(); c.MODEL = CN(); c.INPUT = CN(); c.SOLVER = CN(); c.DATASETS = CN(); c.TEST = CN()
        c.MODEL.BACKBONE = CN(); c.MODEL.FPN = CN(); c.MODEL.RESNETS = CN()
        c.MODEL.DEVICE = "cuda"; c.MODEL.PIXEL_MEAN = [123.675, 116.280, 103.530]; c.MODEL.PIXEL_STD = [58.395, 57.120, 57.375]
        c.MODEL.BACKBONE.NAME = "build_resnet_backbone"; c.MODEL.RESNETS.DEPTH = 50
        c.INPUT.MIN_SIZE_TEST = 800; c.INPUT.MAX_SIZE_TEST = 1333; c.TEST.DETECTIONS_PER_IMAGE = 100
        c.DATASETS.TEST = ("coco_2017_val",)
        return c

    # configs/R50_coco.yaml + R50_ovis_360.yaml on top of add_mdqe_config's defaults
    cfg = base_cfg(); add_mdqe_config(cfg)
    assert cfg.MODEL.MDQE.NUM_OBJECT_QUERIES == 200 and cfg.MODEL.MDQE.APPLY_CLS_THRES == 0.05 and cfg.MODEL.MDQE.MERGE_ON_CPU is False
    cfg.MODEL.MDQE.NUM_CLASSES = 25; cfg.MODEL.MDQE.SAMPLING_FRAME_NUM_TEST = 4; cfg.MODEL.MDQE.MAX_NUM_INSTANCES = 120
    cfg.MODEL.MDQE.WINDOW_FRAME_NUM_TEST = 30; cfg.MODEL.MDQE.APPLY_CLS_THRES = 0.1
    cfg.INPUT.SAMPLING_FRAME_NUM = 4; cfg.INPUT.MIN_SIZE_TEST = 360; cfg.TEST.DETECTIONS_PER_IMAGE = 15
    cfg.DATASETS.TEST = ("ytvis_ovis_val",)
    m = MDQE(cfg)
    want = PRESETS["R50_ovis_360"]
    import dataclasses
    diff = {f.name: (getattr(m.cfg, f.name), getattr(want, f.name)) for f in dataclasses.fields(want) if getattr(m.cfg, f.name) != getattr(want, f.name)}
    assert not diff, diff
    assert m.cfg.is_coco is False and any(n.startswith("detr.backbone.0.backbone.res5") for n, _ in m.named_parameters())

    # configs/swinl_coco.yaml + swinl_ovis.yaml
    cfg = base_cfg(); add_mdqe_config(cfg); add_swinl_config(cfg)
    assert cfg.MODEL.SWIN.WINDOW_SIZE == 24 and cfg.MODEL.FPN.TOP_LEVELS == 2            # mdqe/backbone/config.py:60-75
    cfg.MODEL.BACKBONE.NAME = "build_swinv2_backbone"; cfg.MODEL.SWIN.WINDOW_SIZE = 12
    cfg.MODEL.MDQE.HIDDEN_DIM = 192; cfg.MODEL.MDQE.NUM_CLASSES = 25; cfg.MODEL.MDQE.SAMPLING_FRAME_NUM_TEST = 2
    cfg.MODEL.MDQE.MAX_NUM_INSTANCES = 120; cfg.MODEL.MDQE.WINDOW_FRAME_NUM_TEST = 20; cfg.MODEL.MDQE.APPLY_CLS_THRES = 0.1
    cfg.MODEL.MDQE.MERGE_ON_CPU = True
    cfg.INPUT.SAMPLING_FRAME_NUM = 2; cfg.INPUT.MIN_SIZE_TEST = 480; cfg.TEST.DETECTIONS_PER_IMAGE = 15
    cfg.DATASETS.TEST = ("ytvis_ovis_val",)
    from mdqe_cvpr2023_amd.config import from_d2_cfg
    c = from_d2_cfg(cfg)
    want = PRESETS["swinl_ovis"]
    diff = {f.name: (getattr(c, f.name), getattr(want, f.name)) for f in dataclasses.fields(want) if getattr(c, f.name) != getattr(want, f.name)}
    assert not diff, diff
    print("D2_COMPAT_OK")
''')


def test_d2_registration_and_config_functions_execute():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "D2_COMPAT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
