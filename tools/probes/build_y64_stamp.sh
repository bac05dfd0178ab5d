# builds aicity_action_amd/lib/y64_stamp.so = the kernel library with attention_bwd_w64.hip compiled -DY_STAMP (per-phase s_memtime stamps, results
# invalid); GPU box: MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/y64_stamp.so Y_STAMP=1 python3 tools/opbench.py attnbwd ...
cd "$(dirname "$0")/../aicity_action_amd/csrc" && make -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-inline-asm -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -DY_STAMP ${Y_EXTRA} -c attention_bwd_w64.hip -o /tmp/y64_stamp.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/y64_stamp.so /tmp/y64_stamp.o $(ls ../lib/obj/*.o | grep -v "/attention_bwd_w64.o")
