#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03i; mkdir -p $out
step 300 $out/tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -x
tail -3 $out/tests.log
step 200 $out/bench.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
step 120 $out/heads_new.log python tools/bench_heads.py 20
grep -v amdgpu $out/heads_new.log
export ARTSPEECH_DIAG_LIB=1
step 200 $out/bench_notok.json env AS_NO_GRU_TOKSUM=1 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
python - <<'PY'
import json
for n in ("bench", "bench_notok"):
    d = json.loads(open(f"gpurun_out/r03i/{n}.json").read().strip().splitlines()[-1])
    k = d["kernels_us_per_step"]
    print(n, d["ms_per_step"], d["loss"], {p: v["us_per_step"] for p, v in k.items() if p.startswith(("head", "gru.bwd", "grub"))})
PY
