# timing ablations of the pipelined attention forward (libs built with -DATT_ABL=<bits>; see attention.hip)
# bits: 1 no exp arithmetic, 2 no in-loop DMA, 4 no barrier, 8 no S MFMAs, 16 no PV MFMAs, 32 no LDS reads, 64 no lgkmcnt waits, 128 no vmcnt waits
export MVIT_ATT_PIPE=1
for a in ${ABLS:-0 1 2 4 8 16 32 3 7 24 56 63}; do
  export MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/abl_$a.so
  echo "abl=$a: $(python3 tools/opbench.py attn 8 1 100353 1569 5 2>&1 | tail -1)"
done
