#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 300 $O/smoke.log python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
tail -2 $O/smoke.log
step 400 $O/driver_style.json python3 bench.py --gpus 1 --steps 20 --warmup 5
python3 - <<'PY'
import json
l=[x for x in open("gpurun_out/driver_style.json") if x.startswith('{')][-1]
d=json.loads(l)
print('driver-style', d['ms_per_step'], d['value'], d['exact_fp32']['ms_per_step'], d['transformer_c4']['ms_per_step'])
PY
