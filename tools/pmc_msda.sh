#!/bin/bash
# rocprofv3 --pmc passes over the fused encoder MSDA (one counter group per run; no tracing domains besides --kernel-trace).
# usage (GPU box, from the repo root): bash tools/pmc_msda.sh <out_dir> <tag>
out=${1:-gpurun_out/pmc_msda}; tag=${2:-r02}
mkdir -p "$out"
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" \
           "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TA_TOTAL_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$root/$out/p$i" -o "${tag}_msda_p$i" -- python3 "$root/tools/pmc_msda.py" > "$root/$out/p$i.log" 2>&1
  f=$(find "$root/$out/p$i" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then echo "== $grp"; python3 "$root/tools/pmc_summary.py" "$f" | grep -A12 msda_fused; fi
done
