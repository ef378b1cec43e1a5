#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03g; mkdir -p $out
step 400 $out/tests.log python -m pytest tests -m gpu -q -x
tail -5 $out/tests.log
step 200 $out/bench.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
step 120 $out/metrics_kernels.log python tools/bench_metrics_kernels.py 50
grep -v amdgpu $out/metrics_kernels.log
python - <<'PY'
import json
for n in ["bench"]:
    d = json.loads(open(f"gpurun_out/r03g/{n}.json").read().strip().splitlines()[-1])
    k = d["kernels_us_per_step"]
    print(n, d["ms_per_step"], d["loss"], {p: v["us_per_step"] for p, v in k.items()})
PY
