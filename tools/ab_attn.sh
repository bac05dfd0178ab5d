for w in 4 8; do
  export MVIT_ATT_WAVES=$w
  echo "waves=$w: $(python3 tools/opbench.py attn 8 4 6272 1568 20 2>&1 | tail -1)  |  $(python3 tools/opbench.py attn 8 2 25088 1568 10 2>&1 | tail -1) | $(python3 tools/opbench.py attn 4 4 6272 1568 20 2>&1 | tail -1)"
done
