for pp in 0 1; do
for shape in "802816 384 96 bg" "802816 384 96 b" "802816 288 96 b" "802816 576 96 b" "802816 96 96 br" "802816 96 384 br" "200704 768 192 bg" "200704 192 192 br" "200704 192 768 br" "200704 576 192 b"; do
  echo "pp=$pp: $(MVIT_GEMM_PP=$pp python3 tools/opbench.py gemm $shape 20 2>&1 | tail -1)"
done; done
for mode in pre der; do echo "dual: $(python3 tools/opbench.py gemmdual 802816 384 96 $mode 20 2>&1 | tail -1)"; echo "dual MVIT_GEMM_K96=0: $(MVIT_GEMM_K96=0 python3 tools/opbench.py gemmdual 802816 384 96 $mode 20 2>&1 | tail -1)"; done
