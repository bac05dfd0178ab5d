root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for i in 1 2; do
for epi in b br; do for k in 384 1536; do for m in 50176 32768 65536; do python tools/opbench.py gemm $m 384 $k $epi 30; done; done; done
done > gpurun_out/r2_t21_gemm_tail.txt 2>&1
