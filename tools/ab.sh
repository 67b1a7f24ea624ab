#!/bin/bash
# Same-box A/B of two library builds: tools/ab.sh <old.so> [kbench args]   (box-to-box variance is ~10 %)
OLD=$1; shift
for i in 1 2; do
  echo "=== old ($OLD) run $i"; MDQE_HIP_LIB=$OLD python tools/kbench.py "$@" 2>/dev/null | grep -v "^---" | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('%-20s %8.4f ms %7.1f' % (d['case'], d['ms'], d.get('TFLOPs', d.get('GBps_compulsory', 0))))"
  echo "=== new run $i"; python tools/kbench.py "$@" 2>/dev/null | grep -v "^---" | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('%-20s %8.4f ms %7.1f' % (d['case'], d['ms'], d.get('TFLOPs', d.get('GBps_compulsory', 0))))"
done
