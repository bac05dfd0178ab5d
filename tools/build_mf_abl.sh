# builds ablation / stamp variants of the fused block-tail kernel (csrc/mlp_fused.hip) as separate libraries:
#   lib/mf_<tag>.so for tag in stamp nogelu nodma nobar   (results of these builds are INVALID: timing only)
# GPU box:  MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/mf_<tag>.so python3 tools/mlp_fused_abl.py
cd "$(dirname "$0")/../aicity_action_amd/csrc" && make -j8 >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-inline-asm -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans"
OTHERS=$(ls ../lib/obj/*.o | grep -v "/mlp_fused.o")
build() { /opt/rocm/bin/hipcc $FLAGS $2 ${MF_EXTRA} -c mlp_fused.hip -o /tmp/mf_$1.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/mf_$1.so /tmp/mf_$1.o $OTHERS; }
for t in ${MF_TAGS:-stamp nogelu nodma nobar nogelu_nodma}; do
  case $t in
    stamp) build $t "-DMF_STAMP" ;;
    nogelu) build $t "-DMF_ABL_NOGELU" ;;
    nodma) build $t "-DMF_ABL_NODMA" ;;
    nobar) build $t "-DMF_ABL_NOBAR" ;;
    nogelu_nodma) build $t "-DMF_ABL_NOGELU -DMF_ABL_NODMA" ;;
    nord) build $t "-DMF_ABL_NORD" ;;
    mfma_only) build $t "-DMF_ABL_NOGELU -DMF_ABL_NODMA -DMF_ABL_NORD" ;;
    gelu_only) build $t "-DMF_ABL_NODMA -DMF_ABL_NORD" ;;
    stamp_mfma_only) build $t "-DMF_STAMP -DMF_ABL_NOGELU -DMF_ABL_NODMA -DMF_ABL_NORD" ;;
    stamp_gelu_only) build $t "-DMF_STAMP -DMF_ABL_NODMA -DMF_ABL_NORD" ;;
    stamp_nogelu) build $t "-DMF_STAMP -DMF_ABL_NOGELU" ;;
    *) build $t "$MF_DEFS" ;;
  esac &
done
wait
