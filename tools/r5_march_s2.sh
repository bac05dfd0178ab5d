#!/bin/bash
# round 5: march vs tiled forms of the pooling conv on the shapes the model does not yet route to the march kernel
for shape in "8 1 8 112 112 1"; do
  echo "tiled(maxw 56) : $(python3 tools/opbench.py pool $shape 50 2>/dev/null | tail -1)"
  echo "march(maxw 112): $(MVIT_POOL_MARCH_MAXW=112 python3 tools/opbench.py pool $shape 50 2>/dev/null | tail -1)"
done
for shape in "8 4 8 28 28" "8 2 8 56 56"; do
  echo "pair tiled: $(MVIT_POOL_MARCH=0 python3 tools/opbench.py poolkv $shape 50 2>/dev/null | head -2 | tr '\n' ' ')"
  echo "pair march: $(python3 tools/opbench.py poolkv $shape 50 2>/dev/null | head -2 | tr '\n' ' ')"
done
