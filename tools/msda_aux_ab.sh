#!/bin/bash
# Round 6: does a cache policy on the deformable gather's buffer loads (nt = streaming, sc1 = agent scope / past the L1) change the launch
# times?  `build` (here, hipcc cross-compiles): one library per policy under tools/lab/; `run` (GPU box): tools/msda_dec_scaling.py on each.
set -e
cd "$(dirname "$0")/.."
C=mdqe_cvpr2023_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/lab
  for aux in 2 16 18 1; do
    /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-function -Wno-pass-failed -DMSDA_GATHER_AUX=$aux -c $C/msda_fused.hip -o tools/lab/msda_fused_aux$aux.o
    objs=$(ls $C/*.o | grep -v msda_fused.o)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/lab/libmdqe_hip_aux$aux.so $objs tools/lab/msda_fused_aux$aux.o
  done
else
  for aux in 0 2 16 18 1; do
    lib=tools/lab/libmdqe_hip_aux$aux.so; [ $aux = 0 ] && lib=$C/libmdqe_hip.so
    echo "== gather loads aux=$aux ($lib)"
    MDQE_HIP_LIB=$lib python tools/msda_dec_scaling.py 2>&1 | grep -v amdgpu.ids
  done
fi
