for d in 0 6; do
  export MVIT_ATT_DBG=$d
  echo "dbg=$d: $(python3 tools/opbench.py attn 8 4 6272 1568 20 2>&1 | tail -1)"
done
