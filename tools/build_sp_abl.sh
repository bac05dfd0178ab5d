# builds aicity_action_amd/lib/sp_abl_<bits>.so = the kernel library with skip_pool.hip compiled -DSP_ABL=<bits> (timing ablations,
# results invalid; bits: see skip_pool.hip); usage: tools/build_sp_abl.sh 1 2 4 ...; on the GPU box: MVIT_HIP_LIB=.../sp_abl_N.so tools/opbench.py projpool ...
cd "$(dirname "$0")/../aicity_action_amd/csrc" && make -j8 >/dev/null
for a in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -DSP_ABL=$a -c skip_pool.hip -o /tmp/sp_abl$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/sp_abl_$a.so /tmp/sp_abl$a.o $(ls ../lib/obj/*.o | grep -v "/skip_pool.o") ) &
done
wait
