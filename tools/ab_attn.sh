# A/B of the attention forward variants (environment switches read once by the library)
for v in ${VARIANTS:-"MVIT_ATT_PIPE=0" "MVIT_ATT_PIPE=1" "MVIT_ATT_SLOT=1"}; do
  echo "[$v] $(env $v python3 tools/opbench.py attn 8 4 6273 1569 20 2>&1 | tail -1)  |  $(env $v python3 tools/opbench.py attn 8 2 25089 1569 10 2>&1 | tail -1) | $(env $v python3 tools/opbench.py attn 8 1 100353 1569 5 2>&1 | tail -1) | $(env $v python3 tools/opbench.py attn 8 8 1569 1569 20 2>&1 | tail -1) | $(env $v python3 tools/opbench.py attn 8 2 25089 6273 5 2>&1 | tail -1)"
done
