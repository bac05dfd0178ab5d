#!/bin/bash
# builds the stand-alone probes next to their sources (no GPU needed): tools/probes/build.sh; then on the GPU box: python3 tools/probes/<name>.py
cd "$(dirname "$0")"
for f in store_probe mfma_shape mfma_valu valu_rate store_bw pool_mfma; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1 -o $f.so $f.hip || exit 1
done
