#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3k
mkdir -p $OUT
cd $R
echo "== whole gpu suite + smoke (lean 8-agent kernel as the 2-D default)"
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gputest.log 2>&1; tail -8 $OUT/gputest.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench 1024 / 4096"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $OUT/bench_n1024.json 2> $OUT/bench_n1024.err; python3 -c "
import json;d=json.load(open('$OUT/bench_n1024.json'));print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['admm']['us_per_iteration'], d['row_free_step']['ms_per_step'], d['config']['qp'], d.get('parity_max_abs'))"
timeout -k 10 120 python3 tools/full_solve_timing.py 64 128 1024 > $OUT/full_solve_timing.txt 2>&1; cat $OUT/full_solve_timing.txt
echo "== config 5 unit"
TRIALS=256 bash tools/batch_rate.sh $OUT/batch128_rates.txt "1:1 1:4 4:4 4:5" > /dev/null 2>&1; grep "procs\|rank 0\|all . ranks" $OUT/batch128_rates.txt | cut -c1-100
echo "== done"
