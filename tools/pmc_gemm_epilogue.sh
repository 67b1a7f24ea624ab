#!/bin/bash
# rocprofv3 --pmc passes over the 128 x 128 fp32 GEMM on FFN1's shape, few-instruction epilogue against the general one (FAST_EPI=1 / 0):
# instruction counts by class, matrix-pipe busy, wave cycles.  One counter group per run; --kernel-trace only.
# usage (GPU box, from the repo root): bash tools/pmc_gemm_epilogue.sh <out_dir>
out=${1:-gpurun_out/pmc_gemm_epi}
mkdir -p "$out"
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for fe in 1 0; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_MFMA" \
             "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    FAST_EPI=$fe rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$root/$out/f${fe}_p$i" -o "gemm_epi_f${fe}_p$i" -- python3 "$root/tools/pmc_gemm_epilogue.py" > "$root/$out/f${fe}_p$i.log" 2>&1
    f=$(find "$root/$out/f${fe}_p$i" -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then echo "== FAST_EPI=$fe: $grp"; python3 "$root/tools/pmc_summary.py" "$f" | grep -A12 gemm_nt_f32_k16; else echo "== FAST_EPI=$fe $grp : no output"; tail -3 "$root/$out/f${fe}_p$i.log"; fi
  done
done
