"""Zero-edit drop-in: run a script of the reference with the MI355X `MDQE` behind the name its configs select.

    python -m mdqe_cvpr2023_amd.launch train_net.py --num-gpus 8 --eval-only --config-file configs/R50_ovis_360.yaml MODEL.WEIGHTS w.pth
    python -m mdqe_cvpr2023_amd.launch demo/demo.py --config-file ... --input ... --output ...

Nothing of the reference is edited (north star: "train_net.py --eval-only and demo.py drop in unchanged").  The reference's scripts
import `mdqe` for the config functions, loaders and the evaluator (train_net.py:40-43, demo/demo.py:16), which registers the reference's
own `MDQE` (mdqe/__init__.py:3 -> mdqe/mdqe.py:60-61); `Trainer.build_model(cfg)` (train_net.py:242) then looks "MDQE" up.  This launcher

  1. prepends `mdqe_cvpr2023_amd/_shim` (a `sitecustomize.py`) and this repo's root to PYTHONPATH and sets MDQE_MI355X_AUTOREGISTER=1 --
     in the ENVIRONMENT, so every interpreter the script starts inherits it: detectron2's `launch` (train_net.py:264-271) starts its
     ranks with `torch.multiprocessing.spawn`, whose children re-import `train_net` (not this launcher) in a fresh interpreter;
  2. starts `python script.py <args>` as a CHILD process with that environment and passes its exit code on (signals are forwarded) --
     the script runs exactly as it would from the command line, `__main__` and all; `--in-process` runs it inside this interpreter
     instead (runpy, the hook installed by hand).

The hook imports `mdqe_cvpr2023_amd.meta_arch` right after `mdqe` has been imported, which takes "MDQE" over (the reference's class stays
selectable as "MDQE_REFERENCE"; MDQE_MI355X_REGISTER=alias keeps "MDQE" the reference's and only adds "MDQE_MI355X").  No GPU call is made
here before the script runs."""
import os
import runpy
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(PKG, "_shim")
ROOT = os.path.dirname(PKG)


def prepare_environment(environ=None):
    """Put the shim and the package root on PYTHONPATH (front), set the auto-register flag.  Returns the environment mapping."""
    env = os.environ if environ is None else environ
    parts = [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p]
    for p in (ROOT, SHIM):                             # SHIM ends up first
        if p in parts:
            parts.remove(p)
        parts.insert(0, p)
    env["PYTHONPATH"] = os.pathsep.join(parts)
    env["MDQE_MI355X_AUTOREGISTER"] = "1"
    # the pipeline's streams on 8 hardware queues instead of HIP's default 4 (bench.py: what the sharded schedule's tracker replay needs;
    # neutral for one GPU); an explicit value wins
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    return env


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    in_process = bool(argv) and argv[0] == "--in-process"
    if in_process:
        argv = argv[1:]
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 0 if argv else 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        print("mdqe_cvpr2023_amd.launch: no such script: %s" % argv[0], file=sys.stderr)
        return 2
    if not in_process:
        import signal
        import subprocess
        env = prepare_environment(dict(os.environ))
        # The child is the script's own process tree (detectron2 `launch` / mp.spawn ranks below it).  It stays in the terminal's foreground
        # process group, so a Ctrl-C reaches it ONCE, from the terminal; this parent survives SIGINT (to pass the child's exit code on) and
        # only forwards SIGTERM (which a supervisor sends to this pid alone).  The parent's SIGINT disposition is a no-op HANDLER, not
        # SIG_IGN: an ignored signal stays ignored across fork/exec (and CPython then never installs KeyboardInterrupt), a handler is reset
        # to the default in the child -- the script and its ranks stay interruptible.  Handlers are in place BEFORE the child exists: no
        # window in which a signal kills the parent and orphans the ranks.
        holder = {}
        signal.signal(signal.SIGINT, lambda s, f: None)
        signal.signal(signal.SIGTERM, lambda s, f: holder["p"].send_signal(s) if "p" in holder else sys.exit(128 + s))
        proc = holder["p"] = subprocess.Popen([sys.executable, argv[0]] + argv[1:], env=env)
        rc = proc.wait()
        return 128 - rc if rc < 0 else rc                  # died by signal n: the shell convention 128 + n (sys.exit(-15) would be 241)
    prepare_environment()
    for p in (ROOT, SHIM):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mdqe_mi355x_sitecustomize", os.path.join(SHIM, "sitecustomize.py"))
    hook = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hook)                      # (runs install(); a foreign `sitecustomize` may own the plain module name)
    sys.argv = [argv[0]] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))        # what `python script.py` puts there
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
