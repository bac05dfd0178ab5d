python -m pytest tests/test_hip_train.py tests/test_hip_bwd_ops.py tests/test_hip_ddp.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r2_t9_gpu_tests.log
tools/prof_noside.sh r2_t9_train_noside --mode train > gpurun_out/r2_t9_train_noside.txt 2>&1
rm -rf gpurun_out/r2_t9_train_noside
python bench.py --no-cpu-baseline --no-forward-record > gpurun_out/r2_t9_bench_train.json 2> gpurun_out/r2_t9_bench_train.err
