#!/bin/bash
source tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03j; mkdir -p $out
cd $R
step 400 $out/tests.log python -m pytest tests -m gpu -q -x
tail -3 $out/tests.log
step 300 $out/bench_gpus2.json python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline
tail -c 1500 $out/bench_gpus2.json
cd /tmp && export TMPDIR=/tmp
step 200 $out/pmc1.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc1 -- python3 $R/tools/bench_heads.py 5
step 200 $out/pmc2.log rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/pmc2 -- python3 $R/tools/bench_heads.py 5
cd $R
for d in pmc1 pmc2; do python3 tools/pmc_by_kernel.py $(ls -t $out/$d/*/*counter_collection.csv | head -1) | grep -i "lin_\|wgrad_f32\|gemm_f32" ; done
