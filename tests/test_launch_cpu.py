"""The zero-edit drop-in (`python -m mdqe_cvpr2023_amd.launch <reference script> ...`): an UNEDITED script that only
imports the reference's `mdqe` package ends up building this package's `MDQE` -- in the launched process and in the processes IT starts
with the `spawn` method (detectron2's `launch`, train_net.py:264-271, starts its ranks through torch.multiprocessing.spawn: fresh
interpreters that re-import `train_net`, never the launcher).  detectron2 / fvcore / the reference's `mdqe` are stand-in packages on
disk (absent from this image); the registry asserts on duplicates like fvcore's."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

D2_INIT = ""
D2_CONFIG = textwrap.dedent('''
    class CfgNode(dict):
        def __getattr__(self, k):
            try: return self[k]
            except KeyError: raise AttributeError(k)
        def __setattr__(self, k, v): self[k] = v
''')
D2_MODELING = textwrap.dedent('''
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
    META_ARCH_REGISTRY = Registry("META_ARCH")
    BACKBONE_REGISTRY = Registry("BACKBONE")
    def build_model(cfg):                             # detectron2/modeling/meta_arch/build.py
        return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
''')
REF_MDQE = textwrap.dedent('''
    from detectron2.modeling import BACKBONE_REGISTRY, META_ARCH_REGISTRY
    @META_ARCH_REGISTRY.register()
    class MDQE:                                       # the reference's torch model (mdqe/mdqe.py:60-61)
        def __init__(self, cfg): self.which = "reference"
    @BACKBONE_REGISTRY.register()
    def build_swinv2_backbone(cfg, input_shape):      # mdqe/backbone/swin_transformer_v2.py:675-702
        return "reference swin"
    def add_mdqe_config(cfg): pass
''')
# the stand-in for train_net.py: UNEDITED -- it never mentions mdqe_cvpr2023_amd
SCRIPT = textwrap.dedent('''
    import multiprocessing as mp, os, sys
    from detectron2.config import CfgNode as CN
    from detectron2.modeling import build_model
    from mdqe import add_mdqe_config                  # train_net.py:40 -- registers the reference's MDQE as a side effect

    def make_cfg():
        from mdqe_cvpr2023_amd import add_mdqe_config as fill
        c = CN(); c.MODEL = CN(); c.INPUT = CN(); c.SOLVER = CN(); c.DATASETS = CN(); c.TEST = CN()
        c.MODEL.BACKBONE = CN(); c.MODEL.FPN = CN(); c.MODEL.RESNETS = CN()
        c.MODEL.DEVICE = "cuda"; c.MODEL.PIXEL_MEAN = [123.675, 116.280, 103.530]; c.MODEL.PIXEL_STD = [58.395, 57.120, 57.375]
        c.MODEL.BACKBONE.NAME = "build_resnet_backbone"; c.MODEL.RESNETS.DEPTH = 50
        c.INPUT.MIN_SIZE_TEST = 360; c.INPUT.MAX_SIZE_TEST = 1333; c.TEST.DETECTIONS_PER_IMAGE = 15
        c.DATASETS.TEST = ("ytvis_ovis_val",)
        fill(c)
        c.MODEL.META_ARCHITECTURE = "MDQE"            # what every config of the reference says
        return c

    def worker(rank, out):                            # a rank as detectron2's launch starts it (spawn: a fresh interpreter)
        m = build_model(make_cfg())
        open(out, "w").write("%s.%s" % (type(m).__module__, type(m).__qualname__))
        from detectron2.modeling import BACKBONE_REGISTRY
        open(out + ".bb", "w").write(BACKBONE_REGISTRY.get("build_swinv2_backbone").__module__)

    if __name__ == "__main__":
        assert sys.argv[1:3] == ["--eval-only", "--num-gpus"], sys.argv      # the script sees its own argv
        out = sys.argv[4]
        m = build_model(make_cfg())
        open(out + ".main", "w").write("%s.%s" % (type(m).__module__, type(m).__qualname__))
        ctx = mp.get_context("spawn")
        ps = [ctx.Process(target=worker, args=(r, out + ".rank%d" % r)) for r in range(2)]
        [p.start() for p in ps]; [p.join() for p in ps]
        sys.exit(max(p.exitcode for p in ps) or 7 * 0)
''')


def _tree(tmp_path):
    for rel, src in (("detectron2/__init__.py", D2_INIT), ("detectron2/config.py", D2_CONFIG), ("detectron2/modeling.py", D2_MODELING),
                     ("mdqe/__init__.py", REF_MDQE), ("train_net.py", SCRIPT)):
        f = tmp_path / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(src)
    return str(tmp_path / "train_net.py")


def _env(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("MDQE_MI355X_REGISTER", "MDQE_MI355X_AUTOREGISTER")}
    env["PYTHONPATH"] = str(tmp_path)                 # where the stand-in detectron2 / mdqe live; the launcher adds the rest
    return env


def _read(tmp_path, name):
    return (tmp_path / name).read_text()


def test_launcher_registers_in_the_script_and_in_its_spawned_ranks(tmp_path):
    script = _tree(tmp_path)
    out = str(tmp_path / "built")
    for mode in ([], ["--in-process"]):
        r = subprocess.run([sys.executable, "-m", "mdqe_cvpr2023_amd.launch"] + mode + [script, "--eval-only", "--num-gpus", "2", out],
                           capture_output=True, text=True, env=_env(tmp_path), cwd=ROOT, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        for name in ("built.main", "built.rank0", "built.rank1"):
            assert _read(tmp_path, name) == "mdqe_cvpr2023_amd.meta_arch.MDQE", (mode, name, _read(tmp_path, name))
            os.remove(str(tmp_path / name))
        # the Swin builder of BACKBONE_REGISTRY is this package's as well, in the spawned ranks too
        assert _read(tmp_path, "built.rank0.bb") == "mdqe_cvpr2023_amd.backbone" == _read(tmp_path, "built.rank1.bb")


def test_without_the_launcher_the_same_script_builds_the_reference(tmp_path):
    """The control: plain `python train_net.py` leaves everything to the reference."""
    script = _tree(tmp_path)
    out = str(tmp_path / "built")
    env = _env(tmp_path)
    env["PYTHONPATH"] += os.pathsep + ROOT             # (make_cfg borrows this package's add_mdqe_config for the key table)
    r = subprocess.run([sys.executable, script, "--eval-only", "--num-gpus", "2", out], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert _read(tmp_path, "built.main") == "mdqe.MDQE" and _read(tmp_path, "built.rank1") == "mdqe.MDQE"
    assert _read(tmp_path, "built.rank1.bb") == "mdqe"


def test_alias_mode_and_exit_code_pass_through(tmp_path):
    script = _tree(tmp_path)
    out = str(tmp_path / "built")
    r = subprocess.run([sys.executable, "-m", "mdqe_cvpr2023_amd.launch", script, "--eval-only", "--num-gpus", "2", out],
                       capture_output=True, text=True, env=dict(_env(tmp_path), MDQE_MI355X_REGISTER="alias"), cwd=ROOT, timeout=900)
    assert r.returncode == 0 and _read(tmp_path, "built.rank0") == "mdqe.MDQE"       # "MDQE" stays the reference's; the alias exists
    bad = tmp_path / "boom.py"
    bad.write_text("import sys; sys.exit(5)")
    r = subprocess.run([sys.executable, "-m", "mdqe_cvpr2023_amd.launch", str(bad)], capture_output=True, text=True, env=_env(tmp_path), cwd=ROOT, timeout=300)
    assert r.returncode == 5
    r = subprocess.run([sys.executable, "-m", "mdqe_cvpr2023_amd.launch", str(tmp_path / "missing.py")], capture_output=True, text=True, env=_env(tmp_path),
                       cwd=ROOT, timeout=300)
    assert r.returncode == 2 and "no such script" in r.stderr


def test_ctrl_c_reaches_the_launched_script(tmp_path):
    """the launcher must not hand an IGNORED SIGINT down to the script (an ignored disposition survives fork/exec and CPython
    then never installs KeyboardInterrupt).  SIGINT to the whole process group, as a terminal's Ctrl-C delivers it: the script sees
    KeyboardInterrupt and leaves with 130; the launcher survives the signal and passes that code on."""
    import signal
    import subprocess
    import time
    script = tmp_path / "sleeper.py"
    script.write_text("import signal, sys, time\n"
                      "print('disposition', signal.getsignal(signal.SIGINT) is signal.default_int_handler, flush=True)\n"
                      "try:\n    time.sleep(60)\nexcept KeyboardInterrupt:\n    print('interrupted', flush=True)\n    sys.exit(130)\n")
    p = subprocess.Popen([sys.executable, "-m", "mdqe_cvpr2023_amd.launch", str(script)], cwd=ROOT, stdout=subprocess.PIPE, text=True,
                         start_new_session=True)
    assert p.stdout.readline().strip() == "disposition True"          # Python's own handler is installed in the child
    time.sleep(0.3)
    os.killpg(p.pid, signal.SIGINT)
    out, _ = p.communicate(timeout=30)
    assert "interrupted" in out and p.returncode == 130
