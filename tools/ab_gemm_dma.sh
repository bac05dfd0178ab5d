for shp in "802816 288 96 b" "802816 96 96 br" "802816 384 96 b" "802816 96 384 br" "802816 576 96 b"; do
  set -- $shp
  echo "shape $1 $2 $3 $4: $(python3 tools/opbench.py gemm $1 $2 $3 $4 10 2>&1 | tail -1)"
done
