# builds aicity_action_amd/lib/pp_abl_<bits>.so = the kernel library with linear_pp.hip compiled -DPP_ABL=<bits> (timing ablations,
# results invalid; bits: see linear_pp.hip); usage: tools/build_pp_abl.sh 1 2 4 8 ...  then on the GPU box: tools/ab_pp_abl.sh
cd "$(dirname "$0")/../aicity_action_amd/csrc" && make -j8 >/dev/null
for a in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -DPP_ABL=$a -c linear_pp.hip -o /tmp/pp_abl$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/pp_abl_$a.so /tmp/pp_abl$a.o $(ls ../lib/obj/*.o | grep -v "/linear_pp.o") ) &
done
wait
