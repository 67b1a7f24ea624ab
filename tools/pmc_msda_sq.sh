#!/bin/bash
# rocprofv3 --pmc passes over the fused encoder MSDA, round 4: where do a wave's cycles go (SQ wait / issue buckets), instruction mix,
# texture-path busy and stall counters.  One counter group per run; --kernel-trace only.
# usage (GPU box, from the repo root): bash tools/pmc_msda_sq.sh <out_dir> <tag>
out=${1:-gpurun_out/pmc_msda_sq}; tag=${2:-r04}
mkdir -p "$out"
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$root/$out/counters_avail.txt" 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVE32_INSTS" \
           "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TA_TOTAL_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$root/$out/p$i" -o "${tag}_msda_sq_p$i" -- python3 "$root/tools/pmc_msda.py" > "$root/$out/p$i.log" 2>&1
  f=$(find "$root/$out/p$i" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then echo "== $grp"; python3 "$root/tools/pmc_summary.py" "$f" | grep -A14 msda_fused; else echo "== $grp : no output"; tail -3 "$root/$out/p$i.log"; fi
done
