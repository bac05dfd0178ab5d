# builds aicity_action_amd/lib/w64_stamp.so = the kernel library with attention_w64.hip compiled -DW_STAMP (per-phase s_memtime stamps,
# results invalid); GPU box: MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/w64_stamp.so MVIT_ATT_W64=1 W_STAMP=1 python3 tools/opbench.py attn ...
cd "$(dirname "$0")/../aicity_action_amd/csrc" && make -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-inline-asm -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -DW_STAMP ${W_EXTRA} -c attention_w64.hip -o /tmp/w64_stamp.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/w64_stamp.so /tmp/w64_stamp.o $(ls ../lib/obj/*.o | grep -v "/attention_w64.o")
