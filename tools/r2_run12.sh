root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
tools/prof_noside.sh r2_t12 --mode train > /dev/null 2>&1
KTRACE_ROWS=400 python3 tools/ktrace.py gpurun_out/r2_t12 7 > gpurun_out/r2_t12_shapes_all.txt
rm -rf gpurun_out/r2_t12
