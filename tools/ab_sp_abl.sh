# timing ablations of the fused skip path (libs built by tools/build_sp_abl.sh); GPU box
for a in ${ABLS:-0 1 2 4 8 16 31}; do
  if [ $a = 0 ]; then unset MVIT_HIP_LIB; else export MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/sp_abl_$a.so; fi
  for shp in "8 8 112 112 96 192" "8 8 28 28 384 768"; do
    echo "abl=$a: $(python3 tools/opbench.py projpool $shp 20 2>&1 | grep 'fwd fused')"
  done
done
unset MVIT_HIP_LIB
python3 tools/opbench.py projpool 8 8 112 112 96 192 20 2>&1 | grep projpool
python3 tools/opbench.py projpool 8 8 56 56 192 384 20 2>&1 | grep projpool
python3 tools/opbench.py projpool 8 8 28 28 384 768 20 2>&1 | grep projpool
