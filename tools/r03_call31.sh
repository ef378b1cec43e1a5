#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 600 $O/gemm_tests.log python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or stream_k" || exit 1
tail -2 $O/gemm_tests.log
grep -q failed $O/gemm_tests.log && exit 1
step 200 $O/gemm_ext_bench.log python tools/bench_gemm_ext.py 10
grep "TF/s" $O/gemm_ext_bench.log
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 3 || exit 1
grep transformer $O/bench_transformer.log
step 300 $O/bench1.json python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras
tail -1 $O/bench1.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['loss'])"
step 900 $O/tests.log python -m pytest tests -m gpu -q || exit 1
tail -2 $O/tests.log
