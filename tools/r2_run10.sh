python -m pytest tests/test_hip_engine.py -q -m gpu -x -k graphed 2>&1 | tail -30 > gpurun_out/r2_t10_graph_test.log
python bench.py --mode loop --graph --no-cpu-baseline --no-kernel-timing > gpurun_out/r2_t10_bench_loop_graph.json 2> gpurun_out/r2_t10_bench_loop_graph.err
python bench.py --mode loop --no-cpu-baseline --no-kernel-timing > gpurun_out/r2_t10_bench_loop.json 2> gpurun_out/r2_t10_bench_loop.err
