#!/bin/bash
# round-3 GPU session e: whole suite on the new defaults (rho interval 50, no-NaN persistent kernels, lean kernel v2), benches
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3e
mkdir -p $OUT
cd $R
echo "== whole gpu suite"
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gputest.log 2>&1; tail -12 $OUT/gputest.log
echo "== step times"
timeout -k 10 200 python3 tools/step_time.py 128x2 1024x2 2048x2 2056x2 4096x2 512x3 > $OUT/step_time.txt 2>&1; cat $OUT/step_time.txt
echo "== phase profile, lean kernel at 4096"
timeout -k 10 120 python3 tools/phase_profile.py 4096 > $OUT/phase_profile_lean_n4096.txt 2>&1; cat $OUT/phase_profile_lean_n4096.txt
echo "== bench 1024 / 4096"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $OUT/bench_n1024.json 2> $OUT/bench_n1024.err; python3 -c "
import json;d=json.load(open('$OUT/bench_n1024.json'));print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['admm']['us_per_iteration'], d['row_free_step']['ms_per_step'], d['config']['qp'], d.get('parity_max_abs'))"
timeout -k 10 300 python3 bench.py --agents 4096 --steps 5 --warmup 1 > $OUT/bench_n4096.json 2> $OUT/bench_n4096.err; python3 -c "
import json;d=json.load(open('$OUT/bench_n4096.json'));print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['admm']['us_per_iteration'], d['row_free_step']['ms_per_step'], d['config']['qp'], d.get('parity_max_abs'))"
echo "== full solves"
timeout -k 10 200 python3 tools/full_solve_timing.py 64 256 1024 4096 > $OUT/full_solve_timing.txt 2>&1; cat $OUT/full_solve_timing.txt
echo "== done"
