root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for shp in "50176 1536 384" "50176 384 1536" "50176 1152 384" "50176 384 384" "200704 768 192" "12544 3072 768"; do
for wgs in 0 256 320 384 448 512 640 768; do
  echo "WGS=$wgs"; MVIT_WGRAD_WGS=$wgs python tools/opbench.py wgrad $shp 30
done
done > gpurun_out/r2_t25_wgrad_wgs.txt 2>&1
