root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
rm -rf gpurun_out/traffic
MARKER=attn_fwd tools/traffic.sh attn_fwd gpurun_out/r4_final1_attn_fwd4_hbm_traffic.json --mode fwd --precision bf16 --streams 2 > /dev/null 2>&1
rm -rf gpurun_out/traffic
cd /tmp && export TMPDIR=/tmp
shp="25088 384"; tag=25088x384
rm -rf $root/gpurun_out/tt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/tt/f -- python3 $root/tools/block_tail_bench.py $shp tail 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/tt/w -- python3 $root/tools/block_tail_bench.py $shp tail 5 > /dev/null 2>&1
python3 - <<PY > $root/gpurun_out/r4_final1_pmc_block_tail_${tag}.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$root/gpurun_out/tt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mlp_fused_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("mvit_block_tail_fwd  M x C = $shp, fp16 build, PMC counters per launch (rocprofv3 --pmc, separate passes; n = launches sampled)")
for k, v in sorted(acc.items()):
    print("%-28s %16.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
f = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]) * 2048.0
w = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"]) * 1024.0
M, C = [int(x) for x in "$shp".split()]
print("HBM bytes per launch: fetch %.1f MB (FETCH_SIZE KiB x 2: gfx950 tallies 128-B requests at 64 B), write %.1f MB; algorithmic %.1f MB (o 16 bit + resid fp32 in, out fp32; weights %.1f MB per CU pass stay in L2)" % (f / 1e6, w / 1e6, M * C * 10 / 1e6, 18 * C * C * 2 / 1e6))
PY
rm -rf $root/gpurun_out/tt
cat $root/gpurun_out/r4_final1_pmc_block_tail_${tag}.txt | tail -2; cat $root/gpurun_out/r4_final1_attn_fwd4_hbm_traffic.json | head -8
