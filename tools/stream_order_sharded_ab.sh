# As stream_order_ab.sh, on the one-rank RCCL rehearsal of the sharded bench (RCCL's stream takes a hardware queue before the model's).
for rep in 1 2; do
for o in - wcxfita wxcfita wcfxita xwcfita wcftia; do
  if [ "$o" = "-" ]; then e=""; else e="MDQE_STREAM_ORDER=$o"; fi
  env $e MDQE_BENCH_FORCE_SHARDED=1 python bench.py --steps 8 --warmup 2 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json
  python - "$o" <<'PY'
import json, sys
d = json.load(open("/tmp/_l.json")); print("sharded 1 rank, stream order %-9s  %.1f fps %.1f ms" % (sys.argv[1], d["value"], d["ms_per_step"]), flush=True)
PY
done; done
