#!/bin/bash
# scratch: one gpurun call of round 4
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
step 500 $O/t_train.log python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q
tail -3 $O/t_train.log
cd /tmp && export TMPDIR=/tmp
F="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile --no-exact"
rm -rf $O/xf_new $O/xw_new $O/xf_old $O/xw_old
step 300 $O/xf_new.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/xf_new -- python3 $R/bench.py $F
step 300 $O/xw_new.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/xw_new -- python3 $R/bench.py $F
export ARTSPEECH_DIAG_LIB=$R/artspeech_amd/libartspeech_hip_diag.so AS_NO_XCD_CHUNKS=1
step 300 $O/xf_old.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/xf_old -- python3 $R/bench.py $F
step 300 $O/xw_old.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/xw_old -- python3 $R/bench.py $F
unset AS_NO_XCD_CHUNKS
cd $R
for t in new old; do
  echo "== $t"
  python3 tools/pmc_kernel_bytes.py $(ls $O/xf_$t/*/*_counter_collection.csv | head -1) FETCH_SIZE "false, false, true, false"
  python3 tools/pmc_kernel_bytes.py $(ls $O/xw_$t/*/*_counter_collection.csv | head -1) WRITE_SIZE "false, false, true, false"
  python3 tools/pmc_kernel_bytes.py $(ls $O/xf_$t/*/*_counter_collection.csv | head -1) FETCH_SIZE "splitk_reduce"
  rm -rf $O/xf_$t $O/xw_$t
done
bash tools/timeline.sh xcd_new
export AS_NO_XCD_CHUNKS=1
bash tools/timeline.sh xcd_old
grep -h "gemm_f32_kernel<64, 64, false" $O/xcd_new_timeline.txt | head -5
echo --
grep -h "gemm_f32_kernel<64, 64, false" $O/xcd_old_timeline.txt | head -5
