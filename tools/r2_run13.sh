root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_hip_train.py tests/test_hip_bwd_ops.py tests/test_hip_engine.py tests/test_hip_ddp.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r2_t13_gpu_tests.log
tools/prof_noside.sh r2_t13 --mode train > /dev/null 2>&1
python3 tools/kstats.py gpurun_out/r2_t13 7 60 > gpurun_out/r2_t13_train_noside.txt
KTRACE_ROWS=400 python3 tools/ktrace.py gpurun_out/r2_t13 7 > gpurun_out/r2_t13_shapes_all.txt
rm -rf gpurun_out/r2_t13
python bench.py --no-cpu-baseline --no-forward-record > gpurun_out/r2_t13_bench_train.json 2> gpurun_out/r2_t13_bench_train.err
