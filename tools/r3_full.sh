#!/bin/bash
# one GPU call: the whole GPU suite, then bench lines with the ping-pong GEMM on (default) and off
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r3_suite.txt; cat gpurun_out/r3_suite.txt
python bench.py --no-cpu-baseline > gpurun_out/r3_bench_pp1.json 2> gpurun_out/r3_bench_pp1.err; tail -1 gpurun_out/r3_bench_pp1.json | cut -c1-1500
MVIT_GEMM_PP=0 python bench.py --no-cpu-baseline > gpurun_out/r3_bench_pp0.json 2> gpurun_out/r3_bench_pp0.err; tail -1 gpurun_out/r3_bench_pp0.json | cut -c1-600
