"""Round 4: the compile-time-level form of the staged gathers (126 VGPRs: one block per CU) against round 3's runtime form (68 VGPRs: two
blocks per CU) INSIDE the pipeline -- the single-GPU bench and the sharded schedule through a one-rank RCCL communicator (decoder on
one stream).  (Measured when the compile-time form was the dispatcher's choice for two staged levels; since then the runtime form is the default and
MDQE_MSDA_VARIANT=1032 = bits 8 (staged) + 1024 selects the compile-time form.)
python tools/msda_form_in_pipeline_ab.py [steps] [reps]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "10"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for r in range(reps):
    for sharded in (False, True):
        for name, env in (("default: runtime level", {}), ("compile-time level", {"MDQE_MSDA_VARIANT": "1032"})):
            e = dict(os.environ, **env)
            if sharded:
                e["MDQE_BENCH_FORCE_SHARDED"] = "1"
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "3", "--no-fast-mode", "--no-cpu-baseline"],
                               capture_output=True, text=True, env=e, cwd=ROOT)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode or not lines:
                print(name, "FAILED rc", p.returncode, p.stderr[-600:], flush=True)
                continue
            d = json.loads(lines[-1])
            m = d.get("roofline_msda", {})
            print("%-8s %-26s rep%d  %6.1f frames/s  %7.2f ms/step  msda enc %4.0f us  dec box %4.0f us  dec tp %4.0f us"
                  % ("sharded" if sharded else "single", name, r, d["value"], d["ms_per_step"], m.get("avg_launch_us", 0),
                     m.get("decoder_box", {}).get("avg_launch_us", 0), m.get("decoder_temporal", {}).get("avg_launch_us", 0)), flush=True)
