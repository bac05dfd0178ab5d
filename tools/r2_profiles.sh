#!/bin/bash
# GPU box, repo root: tools/r2_profiles.sh <prefix>   -> gpurun_out/<prefix>_*  (the round's evidence set; copy into profiles/)
pre=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python bench.py > gpurun_out/${pre}_bench_train_bs8_448.json 2> gpurun_out/${pre}_bench_train.err
python bench.py --mode loop --no-cpu-baseline > gpurun_out/${pre}_bench_loop.json 2> gpurun_out/${pre}_bench_loop.err
python bench.py --mode loop --graph --no-cpu-baseline > gpurun_out/${pre}_bench_loop_graph.json 2> gpurun_out/${pre}_bench_loop_graph.err
python bench.py --mode fwd --no-cpu-baseline > gpurun_out/${pre}_bench_fwd_bs8_448.json 2> gpurun_out/${pre}_bench_fwd.err
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
MARKER=attn_fwd tools/traffic.sh attn_fwd gpurun_out/${pre}_attn_fwd4_hbm_traffic.json --mode fwd > /dev/null 2>&1
rm -rf gpurun_out/traffic
MARKER=attn_fwd tools/traffic.sh attn_fwd gpurun_out/${pre}_attn_fwd8_hbm_traffic.json --mode fwd --streams 1 > /dev/null 2>&1
rm -rf gpurun_out/traffic
MARKER=attn_bwd_delta tools/traffic.sh attn_bwd gpurun_out/${pre}_attn_bwd_hbm_traffic.json --mode train > /dev/null 2>&1
rm -rf gpurun_out/traffic
# where the copyBuffer launches fall (initialisation vs per-step)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/${pre}_when -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-forward-record > /dev/null 2>&1
cd $root
(python3 tools/ktrace_when.py gpurun_out/${pre}_when copyBuffer; python3 tools/ktrace_when.py gpurun_out/${pre}_when compare_scalar; python3 tools/ktrace_when.py gpurun_out/${pre}_when reduce_partials) > gpurun_out/${pre}_launch_phases.txt 2>&1
rm -rf gpurun_out/${pre}_when
