#!/bin/bash
# HBM traffic of the kernels from PMC counters, two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE)
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes. usage: tools/gpu_pmc.sh <tag>
TAG=${1:-pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $ROOT/$OUT/$c -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/$c.log 2>&1 )
done
python tools/rocpd_pmc.py $(find $OUT -name '*.db') > $OUT/pmc_hbm_traffic.txt 2>&1
find $OUT -name '*.db' -size +20M -delete
cat $OUT/pmc_hbm_traffic.txt
