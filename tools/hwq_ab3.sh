for rep in 1 2; do for q in 4 5 6 8; do
  env GPU_MAX_HW_QUEUES=$q python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json
  python - "$q" <<'PY'
import json, sys
d = json.load(open("/tmp/_l.json")); print("unsharded GPU_MAX_HW_QUEUES=%s  %.1f fps %.1f ms" % (sys.argv[1], d["value"], d["ms_per_step"]), flush=True)
PY
done; done
