#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r03t; mkdir -p $O
step 600 $O/train_tests.log python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q || exit 1
tail -2 $O/train_tests.log
grep -q failed $O/train_tests.log && exit 1
step 300 $O/bench1.json python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras
tail -1 $O/bench1.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['loss'])"
step 300 $O/bench2.json python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras
tail -1 $O/bench2.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['loss'])"
step 120 $O/recurrence_in_step.log python3 tools/recurrence_stamps.py 50
grep -i "ns per\|fwd\|bwd" $O/recurrence_in_step.log | head -12
