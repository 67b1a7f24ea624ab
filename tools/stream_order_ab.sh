# First-use order of the pipeline's streams (= how they are dealt onto the 4 hardware queues): w work/clip, i instance chain, a decode-ahead
# (high priority); c copy, f frame, t tracker (normal); x / X a dummy normal / high-priority stream.  "-" = the natural order.
for o in - wicfta wcftia fwicta wiacft cftwia wficta wifcat wxicfta wicxfta; do
  if [ "$o" = "-" ]; then e=""; else e="MDQE_STREAM_ORDER=$o"; fi
  env $e python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_l.json
  python - "$o" <<'PY'
import json, sys
d = json.load(open("/tmp/_l.json")); print("stream order %-8s  %.1f fps %.1f ms" % (sys.argv[1], d["value"], d["ms_per_step"]), flush=True)
PY
done
