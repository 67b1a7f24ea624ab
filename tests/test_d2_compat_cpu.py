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
    class Registry:                                   # the contract of fvcore.common.registry.Registry (asserts on duplicates)
        def __init__(self, name): self._name, self._obj_map = name, {}
        def _do_register(self, name, obj):
            assert name not in self._obj_map, "An object named '%s' was already registered in '%s' registry!" % (name, self._name)
            self._obj_map[name] = obj
        def register(self, obj=None):
            if obj is None:
                def deco(o):
                    self._do_register(o.__name__, o); return o
                return deco
            self._do_register(obj.__name__, obj)
        def get(self, name):
            ret = self._obj_map.get(name)
            if ret is None: raise KeyError(name)
            return ret
    d2 = types.ModuleType("detectron2"); d2.__path__ = []
    cfgm = types.ModuleType("detectron2.config"); cfgm.CfgNode = CN
    mod = types.ModuleType("detectron2.modeling"); mod.META_ARCH_REGISTRY = Registry("META_ARCH")
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


# ---- coexistence with the reference's own `mdqe` package ---------------------------------------------------------
# fvcore's Registry refuses a second registration of a name (`assert name not in self._obj_map`); the reference registers ITS model as
# "MDQE" when `mdqe` is imported (mdqe/__init__.py:3, mdqe/mdqe.py:60-61) and train_net.py:40-43 / demo/demo.py:16 import `mdqe`.  The
# stand-ins below keep exactly those two behaviours: a registry that asserts on duplicates, and an `mdqe` package whose import registers
# a class named MDQE.  Both import orders must end with `build_model(cfg)` constructing THIS package's class.
COEXIST = textwrap.dedent('''
    import sys, types, logging
    ORDER = sys.argv[1]
    class CN(dict):
        def __getattr__(self, k):
            try: return self[k]
            except KeyError: raise AttributeError(k)
        def __setattr__(self, k, v): self[k] = v
    class Registry:                                   # the contract of fvcore.common.registry.Registry
        def __init__(self, name): self._name, self._obj_map = name, {}
        def _do_register(self, name, obj):
            assert name not in self._obj_map, "An object named '%s' was already registered in '%s' registry!" % (name, self._name)
            self._obj_map[name] = obj
        def register(self, obj=None):
            if obj is None:
                def deco(o):
                    self._do_register(o.__name__, o); return o
                return deco
            self._do_register(obj.__name__, obj)
        def get(self, name):
            ret = self._obj_map.get(name)
            if ret is None: raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
            return ret
    REG = Registry("META_ARCH")
    d2 = types.ModuleType("detectron2"); d2.__path__ = []
    cfgm = types.ModuleType("detectron2.config"); cfgm.CfgNode = CN
    mod = types.ModuleType("detectron2.modeling"); mod.META_ARCH_REGISTRY = REG
    def build_model(cfg):                             # detectron2/modeling/meta_arch/build.py: registry lookup by cfg.MODEL.META_ARCHITECTURE
        return REG.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    mod.build_model = build_model
    sys.modules.update({"detectron2": d2, "detectron2.config": cfgm, "detectron2.modeling": mod})

    def import_reference():                           # what `from mdqe import add_mdqe_config, ...` triggers (train_net.py:40)
        ref = types.ModuleType("mdqe"); ref.__path__ = []
        sub = types.ModuleType("mdqe.mdqe")
        @REG.register()
        class MDQE:                                   # the reference's torch model
            __module__ = "mdqe.mdqe"
            def __init__(self, cfg): self.which = "reference"
        sub.MDQE = MDQE; ref.MDQE = MDQE
        sys.modules.update({"mdqe": ref, "mdqe.mdqe": sub})
        return MDQE

    records = []
    class H(logging.Handler):
        def emit(self, r): records.append(r.getMessage())
    logging.getLogger("mdqe_cvpr2023_amd").addHandler(H())

    if ORDER == "reference_first":
        RefMDQE = import_reference()
        import mdqe_cvpr2023_amd.meta_arch as ours
    elif ORDER == "ours_first":
        import mdqe_cvpr2023_amd.meta_arch as ours
        RefMDQE = import_reference()                  # would trip fvcore's assertion without the guard
    else:                                             # MDQE_MI355X_REGISTER=alias: the reference keeps its name
        RefMDQE = import_reference()
        import mdqe_cvpr2023_amd.meta_arch as ours

    from mdqe_cvpr2023_amd import add_mdqe_config
    def cfg_for(arch):
        c = CN(); c.MODEL = CN(); c.INPUT = CN(); c.SOLVER = CN(); c.DATASETS = CN(); c.TEST = CN()
        c.MODEL.BACKBONE = CN(); c.MODEL.FPN = CN(); c.MODEL.RESNETS = CN()
        c.MODEL.DEVICE = "cuda"; c.MODEL.PIXEL_MEAN = [123.675, 116.280, 103.530]; c.MODEL.PIXEL_STD = [58.395, 57.120, 57.375]
        c.MODEL.BACKBONE.NAME = "build_resnet_backbone"; c.MODEL.RESNETS.DEPTH = 50
        c.INPUT.MIN_SIZE_TEST = 360; c.INPUT.MAX_SIZE_TEST = 1333; c.TEST.DETECTIONS_PER_IMAGE = 15
        c.DATASETS.TEST = ("ytvis_ovis_val",)
        add_mdqe_config(c)
        c.MODEL.META_ARCHITECTURE = arch
        return c

    st = ours.registration_state()
    if ORDER == "alias_only":
        assert st["state"] == "alias only", st
        assert type(build_model(cfg_for("MDQE"))) is RefMDQE               # untouched
        assert type(build_model(cfg_for("MDQE_MI355X"))) is ours.MDQE_MI355X
        assert "MDQE_REFERENCE" not in REG._obj_map
    else:
        m = build_model(cfg_for("MDQE"))              # every config of the reference says META_ARCHITECTURE: "MDQE"
        assert type(m) is ours.MDQE, type(m)
        assert any(n.startswith("detr.transformer_dec.") for n, _ in m.named_parameters())
        assert type(build_model(cfg_for("MDQE_MI355X"))) is ours.MDQE_MI355X and isinstance(build_model(cfg_for("MDQE_MI355X")), ours.MDQE)
        assert build_model(cfg_for("MDQE_REFERENCE")).which == "reference"  # the reference's model stays selectable
        assert REG.get("MDQE_REFERENCE") is RefMDQE
        assert any("MDQE_REFERENCE" in r for r in records), records          # the take-over is logged, never silent
        ours.register_with_detectron2()                                     # idempotent
        assert REG.get("MDQE") is ours.MDQE and REG.get("MDQE_REFERENCE") is RefMDQE
        # the registry still refuses every OTHER duplicate
        try:
            REG.register(ours.MDQE_MI355X); raise SystemExit("duplicate alias accepted")
        except AssertionError:
            pass
    print("COEXIST_OK", ORDER)
''')


def _coexist(order, env_extra=None):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("MDQE_MI355X_REGISTER", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", COEXIST, order], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "COEXIST_OK " + order in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_coexists_with_reference_package_imported_first():
    _coexist("reference_first")


def test_coexists_with_reference_package_imported_after():
    _coexist("ours_first")


def test_alias_only_registration_leaves_the_reference_in_place():
    _coexist("alias_only", {"MDQE_MI355X_REGISTER": "alias"})


def test_a_broken_detectron2_is_not_swallowed():
    """Only a MISSING detectron2 is tolerated at import; anything else propagates (round 2 swallowed every exception here)."""
    script = textwrap.dedent('''
        import sys, types
        d2 = types.ModuleType("detectron2"); d2.__path__ = []
        mod = types.ModuleType("detectron2.modeling")
        class Bad:
            _obj_map = {}
            def register(self, obj=None): raise RuntimeError("registry is broken")
        mod.META_ARCH_REGISTRY = Bad()
        sys.modules.update({"detectron2": d2, "detectron2.modeling": mod})
        try:
            import mdqe_cvpr2023_amd.meta_arch
        except RuntimeError as e:
            assert "broken" in str(e); print("LOUD_OK")
    ''')
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "LOUD_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_a_detectron2_that_fails_to_import_is_a_warning_not_a_crash():
    """detectron2 installed but `detectron2.modeling` raising ImportError / OSError (its compiled `_C` missing on a ROCm box): the
    detectron2-free entry point must survive -- a logged warning + `registration_state()`; `MDQE_MI355X_REGISTER=strict` raises."""
    script = textwrap.dedent('''
        import importlib.abc, importlib.machinery, logging, os, sys, types
        d2 = types.ModuleType("detectron2"); d2.__path__ = []
        sys.modules["detectron2"] = d2
        class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
            def find_spec(self, name, path, target=None):
                return importlib.machinery.ModuleSpec(name, self) if name == "detectron2.modeling" else None
            def create_module(self, spec): return None
            def exec_module(self, module):
                raise (OSError if os.environ.get("KIND") == "os" else ImportError)("libtorch_hip.so: undefined symbol (detectron2._C)")
        sys.meta_path.insert(0, Finder())
        records = []
        class H(logging.Handler):
            def emit(self, r): records.append(r.getMessage())
        logging.getLogger("mdqe_cvpr2023_amd").addHandler(H())
        try:
            import mdqe_cvpr2023_amd.meta_arch as ours
        except (ImportError, OSError) as e:
            print("RAISED", type(e).__name__); raise SystemExit(0)
        st = ours.registration_state()
        assert st["state"] == "detectron2 import failed" and "_C" in st["error"], st
        assert any("does not import" in r for r in records), records
        from mdqe_cvpr2023_amd import MDQE          # the d2-free surface is intact
        print("WARNED_OK")
    ''')
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("MDQE_MI355X_REGISTER", None)
    for kind in ("import", "os"):
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=dict(env, KIND=kind), cwd=ROOT, timeout=600)
        assert r.returncode == 0 and "WARNED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=dict(env, MDQE_MI355X_REGISTER="strict"), cwd=ROOT,
                       timeout=600)
    assert r.returncode == 0 and "RAISED ImportError" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
