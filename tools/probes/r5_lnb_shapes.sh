for shape in "8 1 8 112 112 1" "8 1 8 112 112 2" "8 2 8 56 56 2" "8 4 8 28 28 2" "8 2 8 56 56 1"; do
  for v in 1 0; do
    out=$(MVIT_POOL_LNB_FUSE=$v timeout 120 python3 tools/opbench.py poolbwd $shape 5 2>&1 | tail -1)
    echo "FUSE=$v $shape -> $out"
  done
done
