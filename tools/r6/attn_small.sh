#!/bin/bash
# attention kernels at small batches (compacted rows, keep words, dropout 0.1): which backward kernel wins where
for B in 4 8 16 36 64; do
  for w in 17 16 8; do
    echo -n "B=$B waves=$w: "; ATTN_BWD_WAVES=$w python tools/attn_bench.py $B 228 0.1 30 2>&1 | grep "compacted" | cut -c1-140
  done
done
