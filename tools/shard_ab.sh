for v in "" "MDQE_DEC_TWO_STREAMS=0" "MDQE_DECODE_AHEAD=0" "MDQE_EARLY_MASKS=0" "MDQE_DEC_TWO_STREAMS=0 MDQE_DECODE_AHEAD=0 MDQE_DEC_FUSED=0"; do
  env MDQE_BENCH_FORCE_SHARDED=1 $v python bench.py --steps 8 --warmup 2 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/_l.json")); print("sharded 1 rank [%s]: %.1f fps %.1f ms" % (sys.argv[1], d["value"], d["ms_per_step"]), flush=True)
PY
done
python bench.py --steps 8 --warmup 2 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json; python -c "
import json; d=json.load(open('/tmp/_l.json')); print('unsharded: %.1f fps %.1f ms' % (d['value'], d['ms_per_step']))"
env MDQE_BENCH_FORCE_SHARDED=1 python bench.py --steps 8 --warmup 2 --no-fast-mode --no-cpu-baseline --chunk-rounds uniform --chunk-windows 4 2>/dev/null | tail -1 > /tmp/_l.json; python -c "
import json; d=json.load(open('/tmp/_l.json')); print('sharded, one 120-frame chunk: %.1f fps %.1f ms' % (d['value'], d['ms_per_step']))"
