set -x
python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r2_t2_gpu_tests.log
python bench.py > gpurun_out/r2_t2_bench_default.json 2> gpurun_out/r2_t2_bench_default.err
python bench.py --mode loop --no-cpu-baseline --no-kernel-timing > gpurun_out/r2_t2_bench_loop.json 2> gpurun_out/r2_t2_bench_loop.err
