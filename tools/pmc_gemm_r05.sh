#!/bin/bash
# rocprofv3 --pmc passes (one counter group per run, --kernel-trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass) over the round's two dominant
# fp32 GEMM kernel forms on the largest launch shapes of a 40-frame pass (tools/pmc_gemm_r05.py).  Per-launch means per kernel.
# usage (GPU box, from the repo root): bash tools/pmc_gemm_r05.sh <out_dir>
out=${1:-gpurun_out/pmc_gemm_r05}
mkdir -p "$out"
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$root/$out/p$i" -o "gemm_r05_p$i" -- python3 "$root/tools/pmc_gemm_r05.py" > "$root/$out/p$i.log" 2>&1
  f=$(find "$root/$out/p$i" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then echo "== $grp"; python3 "$root/tools/pmc_summary.py" "$f" | grep -A12 gemm_nt_f32_k16; cp "$f" "$root/$out/r05_pmc_gemm_p$i.csv"; else echo "== $grp : no output"; tail -3 "$root/$out/p$i.log"; fi
done
