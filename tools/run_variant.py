"""python tools/run_variant.py mha=0 wattn=0 gemm=0 -- bench.py [args]: run a script with debug kernel variants set."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd._lib import lib
i = sys.argv.index("--")
for kv in sys.argv[1:i]:
    k, v = kv.split("=")
    {"mha": lib.mdqe_debug_mha_variant, "wattn": lib.mdqe_debug_window_attn_variant, "gemm": lib.mdqe_debug_gemm_variant, "msdaxcd": lib.mdqe_debug_msda_xcd_order}[k](int(v))
sys.argv = sys.argv[i + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
