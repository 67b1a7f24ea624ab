#!/bin/bash
# per-kernel time of the per-clip stage alone -> gpurun_out/clip_stage/  (summary: gpurun_out/clip_stage_summary.txt)
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/clip_stage
export TMPDIR=/tmp
python3 tools/clip_stage_profile.py 10 > gpurun_out/clip_stage_wall.txt 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/clip_stage -o cs -- python3 $R/tools/clip_stage_profile.py 10 > $R/gpurun_out/clip_stage_prof.log 2>&1
cd $R
f=$(find gpurun_out/clip_stage -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P' > gpurun_out/clip_stage_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms, %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%6.2f %%  %5d x %8.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:150]))
P
cat gpurun_out/clip_stage_wall.txt; head -45 gpurun_out/clip_stage_summary.txt
