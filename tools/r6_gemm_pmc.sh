#!/bin/bash
# round 6 (GPU box, repo root): PMC evidence for linear_big_kernel<16-bit, DG = 2> (fc2 data gradient x saved GELU') at the stage-3 shape -> gpurun_out/r6_pmc_gemm_big_dgder.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{ echo "# linear_big_kernel<bf16, false, false, 2>: y = (dy . W) * GELU'(saved), M = 50176, N = 1536, K = 384 (stage 3, B = 8); 59.2 GFLOP; algorithmic bytes 38.5 MB (dy) + 154 MB (saved derivative) + 154 MB (result)"
  echo "# box: $(hostname) $(date -u +%FT%TZ); tree: ${COMMIT:-a5d9f66}; counters averaged per launch (rocprofv3 --pmc, two SQ passes + FETCH_SIZE / WRITE_SIZE passes)"
  python3 tools/opbench.py gemmdual 50176 1536 384 dgder 30 | tail -1
  tools/pmc.sh r6pmc_g linear_big -- gemmdual 50176 1536 384 dgder 20
  cd /tmp && export TMPDIR=/tmp
  rm -rf $root/gpurun_out/tt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/tt/f -- python3 $root/tools/opbench.py gemmdual 50176 1536 384 dgder 20 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/tt/w -- python3 $root/tools/opbench.py gemmdual 50176 1536 384 dgder 20 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$root/gpurun_out/tt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_big" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc["FETCH_SIZE"] and acc["WRITE_SIZE"]:
    f = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]) * 2048.0
    w = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"]) * 1024.0
    print("HBM-side bytes per launch: fetch %.1f MB (FETCH_SIZE KiB x 2: gfx950 tallies 128-B requests at 64 B), write %.1f MB; algorithmic 192.7 MB in + 154.1 MB out" % (f / 1e6, w / 1e6))
PY
  rm -rf $root/gpurun_out/tt $root/gpurun_out/r6pmc_g; } > $root/gpurun_out/r6_pmc_gemm_big_dgder.txt 2>&1
cat $root/gpurun_out/r6_pmc_gemm_big_dgder.txt
