#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3f
mkdir -p $OUT
cd $R
echo "== whole gpu suite + smoke"
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gputest.log 2>&1; tail -8 $OUT/gputest.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== config 5 unit with more workers than 16 (persistent launches claim their CUs)"
TRIALS=256 bash tools/batch_rate.sh $OUT/batch128_more_workers.txt "4:4 4:5 4:6 5:6 5:8" > /dev/null 2>&1; grep "procs\|all . ranks\|errors" $OUT/batch128_more_workers.txt
echo "== done"
