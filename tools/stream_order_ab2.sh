for rep in 1 2; do
for o in - wficxxxtxxxa wficxxta wcfixta; do
  if [ "$o" = "-" ]; then e=""; else e="MDQE_STREAM_ORDER=$o"; fi
  env $e python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json
  python - "$o" <<'PY'
import json, sys
d = json.load(open("/tmp/_l.json")); print("stream order %-14s  %.1f fps %.1f ms" % (sys.argv[1], d["value"], d["ms_per_step"]), flush=True)
PY
done; done
