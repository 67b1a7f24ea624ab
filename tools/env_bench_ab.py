"""One bench process per setting of an environment variable, alternated (runtime knobs that must be set before HIP initialises).
python tools/env_bench_ab.py VAR "v1 v2 ..." [steps] [reps]     (the value `-` = unset)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, vals = sys.argv[1], sys.argv[2].split()
steps = sys.argv[3] if len(sys.argv) > 3 else "10"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
for r in range(reps):
    for v in vals:
        e = {k: x for k, x in os.environ.items() if k != var}
        if v != "-":
            e[var] = v
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "3", "--no-fast-mode", "--no-cpu-baseline"],
                           capture_output=True, text=True, env=e, cwd=ROOT)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode or not lines:
            print("%s=%s FAILED rc %d %s" % (var, v, p.returncode, p.stderr[-300:]), flush=True)
            continue
        d = json.loads(lines[-1])
        print("%s=%-6s rep%d  %6.1f frames/s  %7.2f ms/step" % (var, v, r, d["value"], d["ms_per_step"]), flush=True)
