for rep in 1 2 3; do for t in 1 0; do
  env MDQE_STREAM_TOUCH=$t python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json
  python - "$t" <<'PY'
import json, sys
d = json.load(open("/tmp/_l.json")); print("MDQE_STREAM_TOUCH=%s unsharded  %.1f fps %.1f ms" % (sys.argv[1], d["value"], d["ms_per_step"]), flush=True)
PY
done; done
