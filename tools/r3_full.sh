#!/bin/bash
# one GPU call: the whole GPU suite (verbose error lines kept), then the default bench line
mkdir -p gpurun_out
(timeout 3000 python -m pytest tests -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -150) > gpurun_out/r3_suite.txt; tail -12 gpurun_out/r3_suite.txt
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; tail -1 gpurun_out/r3_bench.json | python3 -c '
import sys, json
d = json.loads(sys.stdin.read())
print(d["value"], d["ms_per_step"], d["model_roofline"])
print(json.dumps(d.get("forward"))[:1800])
print(json.dumps(d.get("window")))
print(json.dumps(d.get("cpu_baseline")))'
