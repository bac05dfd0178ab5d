# timing ablations of the ping-pong GEMM (libs built by tools/build_pp_abl.sh)
export MVIT_GEMM_PP=1
for a in ${ABLS:-0 1 2 4 16 3 5 6 7}; do
  if [ $a = 0 ]; then unset MVIT_HIP_LIB; else export MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/pp_abl_$a.so; fi
  for shp in "${SHAPES[@]:-50176 384 1536 b}"; do
    echo "abl=$a: $(python3 tools/opbench.py gemm $shp 20 2>&1 | tail -1)"
  done
done
