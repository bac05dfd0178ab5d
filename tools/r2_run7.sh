python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r2_t7_gpu_tests.log
python bench.py --no-cpu-baseline --no-forward-record > gpurun_out/r2_t7_bench_train.json 2> gpurun_out/r2_t7_bench_train.err
