root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_hip_train.py tests/test_hip_bwd_ops.py tests/test_hip_engine.py tests/test_hip_ddp.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r2_t16_gpu_tests.log
for i in 1 2; do
MVIT_LN_EMIT16=0 python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing > gpurun_out/r2_t16_bench_e0_$i.json 2> gpurun_out/r2_t16_bench.err
MVIT_LN_EMIT16=1 python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing > gpurun_out/r2_t16_bench_e1_$i.json 2>> gpurun_out/r2_t16_bench.err
done
tools/prof_noside.sh r2_t16 --mode train > /dev/null 2>&1
python3 tools/kstats.py gpurun_out/r2_t16 7 60 > gpurun_out/r2_t16_train_noside.txt
rm -rf gpurun_out/r2_t16
