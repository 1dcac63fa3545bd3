#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3i
mkdir -p $OUT
cd $R
B=tools/bin/launch_rate
{
echo "== one process, T threads (one stream each), empty kernel / 5 us kernel / 16-workgroup 20 us kernel"
for T in 1 4 8 16; do GPU_MAX_HW_QUEUES=24 $B $T 4000 1 0; done
for T in 1 4 8 16; do GPU_MAX_HW_QUEUES=24 $B $T 4000 1 5; done
for T in 1 4 8 16; do GPU_MAX_HW_QUEUES=24 $B $T 2000 16 20; done
echo "== default HW queues (4)"
for T in 4 16; do $B $T 4000 1 5; done
echo "== four processes x 4 threads at once (5 us kernel)"
for p in 1 2 3 4; do GPU_MAX_HW_QUEUES=8 $B 4 4000 1 5 & done; wait
echo "== four processes x 5 threads at once (16 workgroups, 20 us)"
for p in 1 2 3 4; do GPU_MAX_HW_QUEUES=8 $B 5 2000 16 20 & done; wait
} > $OUT/launch_rate.txt 2>&1
cat $OUT/launch_rate.txt
