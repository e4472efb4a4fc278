#!/bin/bash
out=gpurun_out/r03m; mkdir -p $out
for m in 10000 30000; do for x in 0 1 2; do
  echo "=== m=$m XMODE=$x" >> $out/xmode.txt
  timeout 120 tools/_bin/slice_tune_x$x $m 0.105 0.05 100 >> $out/xmode.txt 2>&1
done; done
grep -n "XMODE\|rows-ascending VT\|greedy-per-group VT\|MISMATCH" $out/xmode.txt | cut -c1-200
