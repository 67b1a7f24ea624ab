"""Round 4: the decoder's LDS-staged gathers (68-72 KB blocks) wait for the frame stream's 32-KB GEMM blocks to retire
and run 2-3x longer inside the pipeline than alone.  Same box, one bench process per setting (--no-fast-mode --no-cpu-baseline), alternated:
  base          the default
  dec24         the decoder's box-level launch stages the coarsest level only (<= 24 KB + descriptors)
  dec24_tp0     ... and the temporal launch takes the gather form (no staging)
  pad12         every K-step-16 GEMM block asks for 12 KB more LDS: 3 GEMM blocks per CU instead of 4
  pad12_dec24   both
python tools/dec_in_pipeline_ab.py [steps] [reps]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "10"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
CFG = (("base", {}), ("dec24", {"MDQE_MSDA_DEC_STAGE_KB": "24"}), ("dec24_tp0", {"MDQE_MSDA_DEC_STAGE_KB": "24", "MDQE_MSDA_TP_STAGED": "0"}),
       ("pad12", {"MDQE_GEMM_LDS_PAD": "12288"}), ("pad12_dec24", {"MDQE_GEMM_LDS_PAD": "12288", "MDQE_MSDA_DEC_STAGE_KB": "24"}))
for r in range(reps):
    for name, env in CFG:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "3", "--no-fast-mode", "--no-cpu-baseline"],
                           capture_output=True, text=True, env=dict(os.environ, **env), cwd=ROOT)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode or not lines:
            print(name, "rep", r, "FAILED rc", p.returncode, p.stderr[-600:], flush=True)
            continue
        d = json.loads(lines[-1])
        m = d["roofline_msda"]
        print("%-12s rep%d  %6.1f frames/s  %7.2f ms/step  gemm %5.1f TF  msda enc %4.0f us  dec box %4.0f us  dec tp %4.0f us"
              % (name, r, d["value"], d["ms_per_step"], d["roofline"]["achieved"], m["avg_launch_us"], m["decoder_box"]["avg_launch_us"],
                 m["decoder_temporal"]["avg_launch_us"]), flush=True)
