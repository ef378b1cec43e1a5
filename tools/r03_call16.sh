#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03p; mkdir -p $out
step 300 $out/tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -x
tail -3 $out/tests.log
step 120 $out/heads.log python tools/bench_heads.py 20
grep -h 'gemm1\|gemm2\|gemm3\|dx3\|dx2\|sum' $out/heads.log
step 200 $out/bench.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
python - <<'PY'
import json
for n in ("bench",):
    d = json.loads([l for l in open(f"gpurun_out/r03p/{n}.json").read().strip().splitlines() if l.startswith("{")][-1])
    k = d["kernels_us_per_step"]
    print(n, d["ms_per_step"], d["loss"], {p: v["us_per_step"] for p, v in k.items() if p.startswith("head")})
PY
