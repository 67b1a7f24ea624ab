#!/bin/bash
# Same-box A/B of an environment switch on the bench: tools/env_ab.sh VAR "v1 v2" [rounds] [bench args...]
VAR=$1; VALS=$2; ROUNDS=${3:-2}; shift 3
for i in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 4 --warmup 1 --no-fast-mode "$@" 2>/dev/null | tail -1 > /tmp/_ab_line.json
    python - "$VAR" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/_ab_line.json"))
print("%s=%s  %.1f fps  %.2f ms/step  frac %.3f" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
  done
done
