# builds aicity_action_amd/lib/stem_abl_<bits>.so = the kernel library with stem.hip compiled -DSTEM_ABL=<bits> (timing ablations, results invalid;
# bits: 1 no halo fill, 2 no MFMA loop, 4 no epilogue stores); usage: tools/build_stem_abl.sh 1 2 4 ...; GPU box: MVIT_HIP_LIB=.../stem_abl_N.so tools/opbench.py stem 8
cd "$(dirname "$0")/../aicity_action_amd/csrc" && make -j8 >/dev/null
for a in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -DSTEM_ABL=$a -c stem.hip -o /tmp/stem_abl$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/stem_abl_$a.so /tmp/stem_abl$a.o $(ls ../lib/obj/*.o | grep -v "/stem.o") ) &
done
wait
