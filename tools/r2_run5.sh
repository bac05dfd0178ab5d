for m in 0 1; do
export MVIT_POOL_MARCH=$m
bash tools/pmc.sh r2_pmc_pool_m$m pool -- pool 8 4 8 28 28 1 > gpurun_out/r2_pmc_pool_m$m.txt 2>&1
rm -rf gpurun_out/r2_pmc_pool_m$m
done
