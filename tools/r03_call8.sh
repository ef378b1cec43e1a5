#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03h; mkdir -p $out
step 300 $out/tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -x
tail -3 $out/tests.log
step 200 $out/bench.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
step 120 $out/heads_new.log python tools/bench_heads.py 20
export ARTSPEECH_DIAG_LIB=1
step 120 $out/heads_old.log env AS_NO_LIN_OUT=1 python tools/bench_heads.py 20
grep -h "gemm3\|sum" $out/heads_new.log $out/heads_old.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03h/bench.json").read().strip().splitlines()[-1])
k = d["kernels_us_per_step"]
print(d["ms_per_step"], d["loss"], {p: v["us_per_step"] for p, v in k.items() if p.startswith("head.")})
PY
