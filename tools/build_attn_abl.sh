# builds aicity_action_amd/lib/abl_<bits>.so = the kernel library with attention.hip compiled -DATT_ABL=<bits> (timing ablations,
# results invalid); usage: tools/build_attn_abl.sh 0 1 2 24 ...   then on the GPU box: ABLS="0 1 2 24" bash tools/ab_attn_abl.sh
cd "$(dirname "$0")/../aicity_action_amd/csrc" && make -j8 >/dev/null
for a in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -DATT_ABL=$a -c attention.hip -o /tmp/attn_abl$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/abl_$a.so /tmp/attn_abl$a.o $(ls ../lib/obj/*.o | grep -v "/attention.o") ) &
done
wait
