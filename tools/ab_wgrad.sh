for shp in "50176 1152 384" "50176 384 384" "50176 1536 384" "50176 384 1536" "200704 768 192" "12544 2304 768" "12544 3072 768" "12544 768 3072"; do
  for e in 0 1; do
    if [ $e = 1 ]; then export MVIT_WGRAD_NO_BIG=1; else unset MVIT_WGRAD_NO_BIG; fi
    echo "shape $shp nobig=$e: $(python3 tools/opbench.py wgrad $shp 20 2>&1 | tail -1)"
  done
done
