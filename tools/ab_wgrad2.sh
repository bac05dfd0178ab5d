for shp in "802816 288 96" "802816 96 96" "802816 384 96" "802816 96 384" "802816 576 96" "200704 192 192" "200704 192 768" "200704 576 192"; do
  for e in 0 1; do
    if [ $e = 1 ]; then export MVIT_WGRAD_NO_BIG=1; else unset MVIT_WGRAD_NO_BIG; fi
    echo "shape $shp nobig=$e: $(python3 tools/opbench.py wgrad $shp 10 2>&1 | tail -1)"
  done
done
