#!/bin/bash
# GPU box, repo root: COMMIT=<short hash> tools/r6_profiles.sh <prefix>   -> gpurun_out/<prefix>_*  (the round's evidence set; copy into profiles/, then python tools/assemble_traffic.py <prefix> <commit> r6)
# The commit is passed in by the caller (COMMIT=$(git rev-parse --short HEAD) on the build container: .git does not travel to the GPU box).
pre=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
echo "box: $(hostname) $(rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -1 | awk '{print $NF}') $(date -u +%FT%TZ) commit ${COMMIT:-unrecorded}" > gpurun_out/${pre}_box.txt
(python -m pytest tests -q -m gpu -s 2>&1 | grep -vE "socket.cpp|^\[Gloo\]" | sed -E "s/^\.+//" | grep -E "\[[a-z0-9_ ,.=]+\]|^ +[0-9]+ +[0-9]+|worst|error|gate|passed|failed|skipped" ) > gpurun_out/${pre}_gputest_verbose.txt
python bench.py > gpurun_out/${pre}_bench_train_bs8_448.json 2> gpurun_out/${pre}_bench_train.err
python bench.py --mode loop --no-cpu-baseline > gpurun_out/${pre}_bench_loop.json 2> /dev/null
python bench.py --mode fwd --no-cpu-baseline > gpurun_out/${pre}_bench_fwd_bs8_448.json 2> gpurun_out/${pre}_bench_fwd.err
python bench.py --mode fwd --precision bf16 --no-cpu-baseline > gpurun_out/${pre}_bench_fwd_bf16_bs8_448.json 2> /dev/null
python bench.py --mode window --no-cpu-baseline > gpurun_out/${pre}_bench_window.json 2> /dev/null
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
# HBM traffic (PMC, separate passes): attention kernels as bench.py times them, and the fused block tail at the stage-3 shape
rm -rf gpurun_out/traffic
MARKER=attn_fwd tools/traffic.sh attn_fwd gpurun_out/${pre}_attn_fwd3_hbm_traffic.json --mode fwd --precision bf16 --batch 9 --streams 3 > /dev/null 2>&1
rm -rf gpurun_out/traffic
MARKER=attn_fwd tools/traffic.sh attn_fwd gpurun_out/${pre}_attn_fwd4_hbm_traffic.json --mode fwd --precision bf16 --streams 2 > /dev/null 2>&1
rm -rf gpurun_out/traffic
MARKER=attn_fwd tools/traffic.sh attn_fwd gpurun_out/${pre}_attn_fwd8_hbm_traffic.json --mode fwd --precision bf16 --streams 1 > /dev/null 2>&1
rm -rf gpurun_out/traffic
MARKER=attn_bwd_dq_kernel tools/traffic.sh attn_bwd gpurun_out/${pre}_attn_bwd_hbm_traffic.json --mode train > /dev/null 2>&1
rm -rf gpurun_out/traffic
cd /tmp && export TMPDIR=/tmp
for shp in "50176 384" "25088 384" "18816 384"; do
  tag=$(echo $shp | tr ' ' 'x')
  rm -rf $root/gpurun_out/tt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/tt/f -- python3 $root/tools/block_tail_bench.py $shp tail 5 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/tt/w -- python3 $root/tools/block_tail_bench.py $shp tail 5 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $root/gpurun_out/tt/p1 -- python3 $root/tools/block_tail_bench.py $shp tail 5 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $root/gpurun_out/tt/p2 -- python3 $root/tools/block_tail_bench.py $shp tail 5 > /dev/null 2>&1
  python3 - <<PY > $root/gpurun_out/${pre}_pmc_block_tail_${tag}.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$root/gpurun_out/tt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mlp_fused_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("mvit_block_tail_fwd  M x C = $shp, fp16 build, PMC counters per launch (rocprofv3 --pmc, separate passes; n = launches sampled)")
for k, v in sorted(acc.items()):
    print("%-28s %16.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
if acc["FETCH_SIZE"] and acc["WRITE_SIZE"]:
    f = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]) * 2048.0
    w = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"]) * 1024.0
    M, C = [int(x) for x in "$shp".split()]
    alg = M * C * (2 + 4 + 4)
    print("HBM bytes per launch: fetch %.1f MB (FETCH_SIZE KiB x 2: gfx950 tallies 128-B requests at 64 B), write %.1f MB; algorithmic %.1f MB (o 16 bit + resid fp32 in, out fp32; weights %.1f MB per CU pass stay in L2)" % (
        f / 1e6, w / 1e6, alg / 1e6, (18 * C * C * 2) / 1e6))
PY
  rm -rf $root/gpurun_out/tt
done
cd $root
(for shp in "50176 384" "25088 384" "18816 384"; do python3 tools/block_tail_bench.py $shp tail 20; python3 tools/block_tail_bench.py $shp mlp 20; done) 2>&1 | grep -v amdgpu > gpurun_out/${pre}_block_tail_alone.txt
(for st in 1 2 3 4; do echo "HIP.STREAMS $st fwd fp16: $(python bench.py --mode fwd --streams $st --no-cpu-baseline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"; done) > gpurun_out/${pre}_fwd_streams.txt 2>&1
# HBM traffic of the two stem kernels (PMC, separate passes)
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/tt
for op in stem stembwd; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/tt/f_$op -- python3 $root/tools/opbench.py $op 8 5 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/tt/w_$op -- python3 $root/tools/opbench.py $op 8 5 > /dev/null 2>&1
done
python3 - <<PY > $root/gpurun_out/${pre}_pmc_stem.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$root/gpurun_out/tt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("stem_ring_kernel", "stem_wgrad_rows_kernel", "stem_pos_bwd_kernel", "slab_sum_kernel"):
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("stem kernels, B = 8 16x448x448 (bf16 build), HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH_SIZE KiB x 2: gfx950")
print("tallies 128-B requests at 64 B); algorithmic: clip 308.3 MB fp32 + tokens 308.3 MB fp32 (forward: read + written; weight gradient: both read)")
for k, d in acc.items():
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) * 2048.0
    w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) * 1024.0
    print("%-26s fetch %8.1f MB   write %8.1f MB   (n = %d / %d launches)" % (k, f / 1e6, w / 1e6, len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"])))
PY
rm -rf $root/gpurun_out/tt
cd $root
