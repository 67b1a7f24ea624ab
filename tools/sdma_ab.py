"""Host<->device copies through the SDMA engines or through blit kernels (`__amd_rocclr_copyBuffer`: 2.3 % of the bench's kernel time, two
thirds of it PCIe-bound blits of the final masks and the frame upload)?  One bench process per setting of HSA_ENABLE_SDMA, alternated.
python tools/sdma_ab.py [steps] [reps]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "10"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for r in range(reps):
    for name, env in (("default (unset)", {}), ("HSA_ENABLE_SDMA=1", {"HSA_ENABLE_SDMA": "1"}), ("HSA_ENABLE_SDMA=0", {"HSA_ENABLE_SDMA": "0"})):
        e = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_SDMA"}
        e.update(env)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "3", "--no-fast-mode", "--no-cpu-baseline"],
                           capture_output=True, text=True, env=e, cwd=ROOT)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode or not lines:
            print(name, "FAILED rc", p.returncode, p.stderr[-400:], flush=True)
            continue
        d = json.loads(lines[-1])
        print("%-20s rep%d  %6.1f frames/s  %7.2f ms/step" % (name, r, d["value"], d["ms_per_step"]), flush=True)
