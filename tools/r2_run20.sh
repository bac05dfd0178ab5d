root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
export MVIT_NO_SIDE_STREAM=1
for i in 1 2; do
for lq in 6272 6144 6400 8192; do python tools/opbench.py attn 8 4 $lq 1569 30; done
done > gpurun_out/r2_t20_attn_tail.txt 2>&1
tools/ab_lib.sh t attnbwd 8 4 6144 1569 30 >> gpurun_out/r2_t20_attn_tail.txt 2>&1
