#!/bin/bash
# round 3, first GPU call: the test suite on the new tree, a short bench line, microbenchmarks that decide the GEMM work.
source tools/gpu_steps.sh
out=gpurun_out/r03a; mkdir -p $out
step 400 $out/tests.log python -m pytest tests -m gpu -x -q
tail -3 $out/tests.log
step 200 $out/bench.json python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras
tail -c 600 $out/bench.json
step 120 $out/metrics_kernels.log python tools/bench_metrics_kernels.py 50 --json $out/metrics_kernels.json
cat $out/metrics_kernels.log
export ARTSPEECH_DIAG_LIB=1
step 120 $out/wgrad_default.log python tools/bench_wgrad.py 20
step 120 $out/wgrad_all.log env AS_WGRAD_MIN_WORK=1 python tools/bench_wgrad.py 20
step 120 $out/wgrad_all_192.log env AS_WGRAD_MIN_WORK=1 python tools/bench_wgrad.py 20 192
cat $out/wgrad_default.log $out/wgrad_all.log $out/wgrad_all_192.log
step 200 $out/bench_diag.json python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras
step 200 $out/bench_dx1lin.json env AS_DX1_LIN=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras
step 200 $out/bench_wgrad_all.json env AS_WGRAD_MIN_WORK=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras
python - <<'PY'
import json
for n in ("bench", "bench_diag", "bench_dx1lin", "bench_wgrad_all"):
    try:
        d = json.loads(open(f"gpurun_out/r03a/{n}.json").read().strip().splitlines()[-1])
        k = d["kernels_us_per_step"]
        print(n, d["ms_per_step"], {p: k[p]["us_per_step"] for p in ("grub.dx1", "grub.dw_hh", "grub.dw_ih1", "trunkb.dw", "gru.bwd_l0", "gru.bwd_l1", "headb.dw2")})
    except Exception as e:
        print(n, "unreadable", e)
PY
echo done
