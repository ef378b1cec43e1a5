#!/bin/bash
# Backward-schedule experiments (AS_SCHED bit mask, csrc/artspeech.hip): ms/step of the default bench for each variant.
# usage (GPU box): bash tools/sched_sweep.sh "0 1 2 3 5 7" > gpurun_out/sched.log
for v in ${1:-0 1 2 3}; do
  AS_SCHED=$v python bench.py --steps 200 --warmup 20 --no-extras --no-profile --no-cpu-baseline > /tmp/sched_$v.json 2>/tmp/sched_$v.err || { echo "AS_SCHED=$v failed"; tail -5 /tmp/sched_$v.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
line = [l for l in open(f"/tmp/sched_{v}.json") if l.startswith("{")][-1]
d = json.loads(line)
print(f"AS_SCHED={v}: {d['ms_per_step']:.4f} ms/step  {d['value']:.0f} frames/s", flush=True)
PY
done
