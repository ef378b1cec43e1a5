#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03n; mkdir -p $out
step 300 $out/tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -x
tail -3 $out/tests.log
step 200 $out/bench.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
step 200 $out/bench2.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
step 120 $out/stamps.log python tools/recurrence_stamps.py 50
grep -v amdgpu $out/stamps.log
python - <<'PY'
import json
for n in ("bench", "bench2"):
    d = json.loads([l for l in open(f"gpurun_out/r03n/{n}.json").read().strip().splitlines() if l.startswith("{")][-1])
    k = d["kernels_us_per_step"]
    print(n, d["ms_per_step"], d["loss"], {p: v["us_per_step"] for p, v in k.items() if p.startswith("gru.")})
PY
