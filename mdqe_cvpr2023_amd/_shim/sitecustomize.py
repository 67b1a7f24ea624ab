"""Interpreter start-up hook of `python -m mdqe_cvpr2023_amd.launch` (this directory rides on PYTHONPATH, so every Python process the
launched script starts -- detectron2's `launch` -> `torch.multiprocessing.spawn` ranks included -- runs it too).

When MDQE_MI355X_AUTOREGISTER=1: the moment the reference's `mdqe` package has finished importing (which registers ITS `MDQE` in
detectron2's META_ARCH_REGISTRY, mdqe/__init__.py:3 -> mdqe/mdqe.py:60-61), `mdqe_cvpr2023_amd.meta_arch` is imported, which takes the
name "MDQE" over (meta_arch.register_with_detectron2).  No file of the reference is edited.  Nothing heavy happens here: the hook only
watches for one module name."""
import importlib.abc
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


class _AfterImport(importlib.abc.MetaPathFinder):
    """Wraps the loader of the top-level module `name`: after its exec_module, `then()` runs once."""

    def __init__(self, name, then):
        self.name, self.then, self.busy = name, then, False

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.name or self.busy:
            return None
        self.busy = True                               # (the real finders answer the nested lookup)
        try:
            spec = importlib.util.find_spec(fullname)
        finally:
            self.busy = False
        if spec is None or spec.loader is None or not hasattr(spec.loader, "exec_module"):
            return spec
        inner, then = spec.loader, self.then

        class Loader(importlib.abc.Loader):
            def create_module(self_, s):
                return inner.create_module(s)

            def exec_module(self_, module):
                inner.exec_module(module)
                then()

            def __getattr__(self_, k):                 # get_code, get_source, is_package ... (runpy, inspect)
                return getattr(inner, k)
        spec.loader = Loader()
        return spec


def _register():
    import mdqe_cvpr2023_amd.meta_arch  # noqa: F401  (registers at import; logs the take-over)


def install():
    if os.environ.get("MDQE_MI355X_AUTOREGISTER") != "1" or getattr(sys, "_mdqe_mi355x_hook", False):
        return
    sys._mdqe_mi355x_hook = True
    if "mdqe" in sys.modules:                          # already imported (install() called late): register now
        _register()
    else:
        sys.meta_path.insert(0, _AfterImport("mdqe", _register))


def _chain():
    """A `sitecustomize` further down sys.path is shadowed by this one: run it too."""
    for p in sys.path:
        try:
            if not p or os.path.abspath(p) == _HERE:
                continue
            f = os.path.join(p, "sitecustomize.py")
            if os.path.isfile(f):
                spec = importlib.util.spec_from_file_location("_mdqe_chained_sitecustomize", f)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                return
        except Exception:                              # (a site hook must never stop the interpreter from starting)
            return


install()
if __name__ == "sitecustomize":
    _chain()
