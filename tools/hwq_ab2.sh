for rep in 1 2; do
for q in 4 8; do
  for two in 1 0; do
    env GPU_MAX_HW_QUEUES=$q MDQE_DEC_TWO_STREAMS=$two MDQE_BENCH_FORCE_SHARDED=1 python bench.py --steps 8 --warmup 2 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json
    python - "$q" "$two" <<'PY'
import json, sys
d = json.load(open("/tmp/_l.json")); print("sharded 1 rank  GPU_MAX_HW_QUEUES=%s  DEC_TWO_STREAMS=%s  %.1f fps %.1f ms" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"]), flush=True)
PY
  done
done
done
