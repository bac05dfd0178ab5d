for shp in "50176 384 384" "50176 384 1536" "200704 192 192" "200704 192 768" "12544 768 768" "12544 768 3072"; do
  echo "shape $shp res: $(python3 tools/opbench.py gemm $shp br 20 2>&1 | tail -1)"
done
