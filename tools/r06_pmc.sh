#!/bin/bash
# Round 6: the eight --pmc passes over the encoder's and the decoder's temporal deformable launch on the round's final library (kernels unchanged
# since round 4: the counters should reproduce r04's) + the GEMM passes.     bash tools/r06_pmc.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
bash tools/pmc_msda.sh gpurun_out/r06/pmc_msda_enc r06 > gpurun_out/r06/r06_pmc_msda_enc_summary.txt 2>&1
MSDA_FORM=temporal bash tools/pmc_msda.sh gpurun_out/r06/pmc_msda_tp r06tp > gpurun_out/r06/r06_pmc_msda_temporal_summary.txt 2>&1
bash tools/pmc_gemm_r05.sh gpurun_out/r06/pmc_gemm > gpurun_out/r06/r06_pmc_gemm_summary.txt 2>&1
find gpurun_out/r06 -type d -name "p[0-9]" -exec rm -rf {} + 2>/dev/null
tail -45 gpurun_out/r06/r06_pmc_msda_enc_summary.txt; tail -20 gpurun_out/r06/r06_pmc_msda_temporal_summary.txt; tail -30 gpurun_out/r06/r06_pmc_gemm_summary.txt
