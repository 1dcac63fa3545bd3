#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3g
mkdir -p $OUT
cd $R
echo "== config 5: naps between polls of the persistent kernels (fabric load of 16 concurrent solves)"
for NAP in 1 4 16; do
  SCP_PERSIST_SPIN_SLEEP=$NAP TRIALS=256 bash tools/batch_rate.sh $OUT/batch128_nap$NAP.txt "4:4 4:5" > /dev/null 2>&1
  echo "nap $NAP: $(grep 'all . ranks' $OUT/batch128_nap$NAP.txt | sed 's/.*= //' | tr '\n' ' ')"
done
echo "== config 5: process x stream layouts"
TRIALS=256 bash tools/batch_rate.sh $OUT/batch128_layouts.txt "1:8 1:12 2:8 3:5 3:6 2:10" > /dev/null 2>&1; grep "procs\|rank 0\|all . ranks" $OUT/batch128_layouts.txt | grep -v "rank [1-9]" | cut -c1-110
echo "== step time 1024 with naps 1 / 4"
for NAP in 1 4; do SCP_PERSIST_SPIN_SLEEP=$NAP timeout -k 10 100 python3 tools/step_time.py 1024x2 4096x2 2>&1 | grep "N="; done
echo "== done"
