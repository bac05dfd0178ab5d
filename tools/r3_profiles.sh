#!/bin/bash
# GPU box, repo root: tools/r3_profiles.sh <prefix>   -> gpurun_out/<prefix>_*  (the round's evidence set; copy into profiles/)
pre=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
(python -m pytest tests -q -m gpu -s 2>&1 | grep -vE "socket.cpp|^\[Gloo\]" | sed -E "s/^\.+//" | grep -E "\[[a-z0-9_ ]+\]|error|gate|passed|failed|skipped" ) > gpurun_out/${pre}_gputest_verbose.txt
python bench.py > gpurun_out/${pre}_bench_train_bs8_448.json 2> gpurun_out/${pre}_bench_train.err
python bench.py --mode loop --no-cpu-baseline > gpurun_out/${pre}_bench_loop.json 2> gpurun_out/${pre}_bench_loop.err
python bench.py --mode loop --graph --no-cpu-baseline > gpurun_out/${pre}_bench_loop_graph.json 2> gpurun_out/${pre}_bench_loop_graph.err
python bench.py --mode fwd --no-cpu-baseline > gpurun_out/${pre}_bench_fwd_bs8_448.json 2> gpurun_out/${pre}_bench_fwd.err
python bench.py --mode fwd --precision bf16 --no-cpu-baseline > gpurun_out/${pre}_bench_fwd_bf16_bs8_448.json 2> /dev/null
python bench.py --mode window --no-cpu-baseline > gpurun_out/${pre}_bench_window.json 2> gpurun_out/${pre}_bench_window.err
# kernel profiles: side streams off (per-kernel durations) and on (as shipped)
for m in train fwd; do
  tools/prof_noside.sh ${pre}_${m}_noside --mode $m > /dev/null 2>&1
  python3 tools/kstats.py gpurun_out/${pre}_${m}_noside 7 80 > gpurun_out/${pre}_${m}_noside_per_step.txt
  python3 tools/ktrace.py gpurun_out/${pre}_${m}_noside 7 > gpurun_out/${pre}_${m}_noside_shapes.txt
  cp $(find gpurun_out/${pre}_${m}_noside -name '*kernel_stats.csv' | head -1) gpurun_out/${pre}_${m}_bs8_448_noside_kernel_stats.csv
  rm -rf gpurun_out/${pre}_${m}_noside
  tools/prof.sh ${pre}_${m}_side --mode $m > /dev/null 2>&1
  python3 tools/kstats.py gpurun_out/${pre}_${m}_side 7 80 > gpurun_out/${pre}_${m}_per_step.txt
  cp $(find gpurun_out/${pre}_${m}_side -name '*kernel_stats.csv' | head -1) gpurun_out/${pre}_${m}_bs8_448_kernel_stats.csv
  rm -rf gpurun_out/${pre}_${m}_side
done
# HBM traffic of the attention kernels (PMC, separate passes)
rm -rf gpurun_out/traffic
# (three sub-batch streams: 8 clips run as 3, 3, 2; --batch 9 makes every launch a 3-clip one, the size bench.py times)
MARKER=attn_fwd tools/traffic.sh attn_fwd gpurun_out/${pre}_attn_fwd3_hbm_traffic.json --mode fwd --precision bf16 --batch 9 > /dev/null 2>&1
rm -rf gpurun_out/traffic
MARKER=attn_fwd tools/traffic.sh attn_fwd gpurun_out/${pre}_attn_fwd8_hbm_traffic.json --mode fwd --precision bf16 --streams 1 > /dev/null 2>&1
rm -rf gpurun_out/traffic
MARKER=attn_bwd_delta tools/traffic.sh attn_bwd gpurun_out/${pre}_attn_bwd_hbm_traffic.json --mode train > /dev/null 2>&1
rm -rf gpurun_out/traffic
# GEMM shapes (this library, ping-pong on / off) and the SQ counters of the ping-pong kernel on the fc1 and qkv shapes
bash tools/r3_gemm_ab.sh gpurun_out/${pre}_gemm_shapes.txt > /dev/null 2>&1
MVIT_GEMM_PP=1 tools/pmc.sh ${pre}_pmc_fc1 linear_pp -- gemm 50176 1536 384 bg 10 > gpurun_out/${pre}_pmc_gemm_pp_fc1.txt 2>&1
MVIT_GEMM_PP=0 tools/pmc.sh ${pre}_pmc_fc1o linear_pers -- gemm 50176 1536 384 bg 10 > gpurun_out/${pre}_pmc_gemm_128x192_fc1.txt 2>&1
MVIT_GEMM_PP=1 tools/pmc.sh ${pre}_pmc_fc2 linear_pp -- gemm 50176 384 1536 b 10 > gpurun_out/${pre}_pmc_gemm_pp_fc2_plain.txt 2>&1
rm -rf gpurun_out/${pre}_pmc_fc1 gpurun_out/${pre}_pmc_fc1o gpurun_out/${pre}_pmc_fc2
# attention forward (64-query kernel): issue counters, kernel alone against the 32-query kernels; skip path op pairs; streams
tools/pmc.sh ${pre}_pmc_w64 attn_fwd_w64 -- attn 8 4 6272 1568 10 > gpurun_out/${pre}_pmc_attn_fwd_w64.txt 2>&1
rm -rf gpurun_out/${pre}_pmc_w64
(for w in 0 1; do for shp in "8 1 100352 1568 10" "8 2 25088 1568 20" "8 4 6272 1568 50" "8 8 1568 1568 50"; do echo "MVIT_ATT_W64=$w: $(MVIT_ATT_W64=$w python3 tools/opbench.py attn $shp 2>&1 | tail -1)"; done; done) > gpurun_out/${pre}_attn_w64_alone.txt 2>&1
(python3 tools/opbench.py projpool 8 8 112 112 96 192 20; python3 tools/opbench.py projpool 8 8 56 56 192 384 20; python3 tools/opbench.py projpool 8 8 28 28 384 768 20) 2>&1 | grep projpool > gpurun_out/${pre}_skip_path_ops.txt
(for st in 1 2 3 4; do echo "HIP.STREAMS $st fwd fp16: $(python bench.py --mode fwd --streams $st --no-cpu-baseline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"; done) > gpurun_out/${pre}_fwd_streams.txt 2>&1
