#!/bin/bash
# Which engine carries a device -> pinned-host copy of a few MB (the per-track mask read-back)?  tools/d2h_probe.py under rocprofv3 --kernel-trace: a blit
# copy shows up as `__amd_rocclr_copyBuffer` launches, an SDMA copy does not.  One run per setting of the runtime's copy knobs.   bash tools/sdma_probe.sh
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for setting in "default" "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_FORCE_BLIT_COPY_SIZE=1048576" "HSA_ENABLE_SDMA=1" "GPU_BLIT_ENGINE_TYPE=1" "GPU_BLIT_ENGINE_TYPE=2"; do
  ( if [ "$setting" != "default" ]; then export "$setting"; fi
    rm -rf /tmp/sdma_probe; rocprofv3 --kernel-trace --stats -d /tmp/sdma_probe -o p -- python3 "$root/tools/d2h_probe.py" > /tmp/sdma_probe.log 2>&1
    db=$(find /tmp/sdma_probe -name "*.db" | head -1)
    n=$(python3 -c "
import sqlite3,sys
c=sqlite3.connect('$db')
r=c.execute(\"select total_calls, total_duration from top_kernels where name like '%copyBuffer%'\").fetchall()
print(r)
" 2>&1)
    echo "$setting: copyBuffer kernels (calls, us): $n | $(grep 'ONE copy' /tmp/sdma_probe.log | cut -c60-140)" )
done
