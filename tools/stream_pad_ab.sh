# MDQE_STREAM_PAD (how the model's normal-priority streams fall onto the 4 hardware queues) on the single-GPU bench and on the one-rank
# RCCL rehearsal of the sharded bench.   bash tools/stream_pad_ab.sh
for mode in "" "MDQE_BENCH_FORCE_SHARDED=1"; do
  for k in 0 1 2 3; do
    env MDQE_STREAM_PAD=$k $mode python bench.py --steps 8 --warmup 2 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json
    python - "$k" "$mode" <<'PY'
import json, sys
d = json.load(open("/tmp/_l.json")); print("MDQE_STREAM_PAD=%s %-28s %.1f fps %.1f ms" % (sys.argv[1], sys.argv[2] or "unsharded", d["value"], d["ms_per_step"]), flush=True)
PY
  done
done
