# HIP streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues (default 4): does the pipeline (frame / clip / instance-chain /
# tracker / copy / decode-ahead streams, + RCCL's in the sharded schedule) run better with more?   bash tools/hwq_ab.sh
for q in 4 8 16; do
  for mode in "" "MDQE_BENCH_FORCE_SHARDED=1"; do
    env GPU_MAX_HW_QUEUES=$q $mode python bench.py --steps 8 --warmup 2 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json
    python - "$q" "$mode" <<'PY'
import json, sys
d = json.load(open("/tmp/_l.json")); print("GPU_MAX_HW_QUEUES=%s %-28s %.1f fps %.1f ms" % (sys.argv[1], sys.argv[2] or "unsharded", d["value"], d["ms_per_step"]), flush=True)
PY
  done
done
