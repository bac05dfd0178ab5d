root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python tools/blaslt_wgrad_ref.py > gpurun_out/r2_blaslt_wgrad_ref.txt 2>&1
for shp in "50176 1536 384" "50176 384 1536" "50176 1152 384" "50176 384 384" "200704 768 192" "200704 576 192" "802816 384 96" "12544 3072 768" "12544 2304 768"; do
  python tools/opbench.py wgrad $shp 20
done > gpurun_out/r2_t19_wgrad_mine.txt 2>&1
