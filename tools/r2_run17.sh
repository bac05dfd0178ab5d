root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_hip_ops.py tests/test_hip_model.py tests/test_hip_train.py tests/test_hip_bwd_ops.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r2_t17_gpu_tests.log
tools/prof_noside.sh r2_t17 --mode train > /dev/null 2>&1
python3 tools/kstats.py gpurun_out/r2_t17 7 60 > gpurun_out/r2_t17_train_noside.txt
rm -rf gpurun_out/r2_t17
python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing > gpurun_out/r2_t17_bench.json 2> gpurun_out/r2_t17_bench.err
python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing > gpurun_out/r2_t17_bench_fwd.json 2>> gpurun_out/r2_t17_bench.err
