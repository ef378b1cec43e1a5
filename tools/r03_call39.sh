#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 300 $O/tri.log timeout -k 10 280 python -m pytest tests/test_gpu_parity.py -x -q -k "triangular or gemm" || exit 1
tail -2 $O/tri.log
grep -q failed $O/tri.log && exit 1
step 300 $O/bench1.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
tail -1 $O/bench1.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['loss'])"
step 300 $O/bench2.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
tail -1 $O/bench2.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['loss'])"
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 5 || exit 1
grep "transformer f" $O/bench_transformer.log
step 200 $O/ge.log python tools/bench_gemm_ext.py 10
grep "attention" $O/ge.log
