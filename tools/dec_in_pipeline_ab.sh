#!/bin/bash
# Round 4: the decoder's LDS-staged gathers (68-72 KB blocks) wait for the frame stream's 32-KB GEMM blocks to retire
# and run 2-3x longer inside the pipeline than alone.  Same box, one bench process per setting (--no-fast-mode --no-cpu-baseline):
#   base            the default
#   dec24           the decoder's box-level launch stages the coarsest level only (<= 24 KB + descriptors)
#   dec24_tp0       ... and the temporal launch takes the gather form (no staging)
#   pad12           every K-step-16 GEMM block asks for 12 KB more LDS: 3 GEMM blocks per CU instead of 4
#   pad12_dec24     both
# usage: bash tools/dec_in_pipeline_ab.sh [steps] [reps]
steps=${1:-10}; reps=${2:-2}
for r in $(seq $reps); do
  for cfg in "base:" "dec24:MDQE_MSDA_DEC_STAGE_KB=24" "dec24_tp0:MDQE_MSDA_DEC_STAGE_KB=24 MDQE_MSDA_TP_STAGED=0" "pad12:MDQE_GEMM_LDS_PAD=12288" \
             "pad12_dec24:MDQE_GEMM_LDS_PAD=12288 MDQE_MSDA_DEC_STAGE_KB=24"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    line=$(env $envs python bench.py --steps $steps --warmup 3 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$name rep$r $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); m=d['roofline_msda']; print('%.1f frames/s  %.2f ms/step  gemm %.1f TF  msda enc %.0f us  dec box %.0f us  dec tp %.0f us' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], m['avg_launch_us'], m['decoder_box']['avg_launch_us'], m['decoder_temporal']['avg_launch_us']))" "$line")"
  done
done
