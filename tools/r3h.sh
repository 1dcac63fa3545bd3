#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3h
mkdir -p $OUT
cd $R
echo "== lean kernel with the two-level all-reduce: tests"
timeout -k 10 400 python -m pytest tests/test_qp_gpu.py tests/test_scp_gpu.py -m gpu -q -k "persistent or rho_switch or lean or config4" > $OUT/lean_tests.log 2>&1; tail -6 $OUT/lean_tests.log
echo "== step time, phase profile"
timeout -k 10 200 python3 tools/step_time.py 2056x2 3000x2 4096x2 > $OUT/step_time.txt 2>&1; cat $OUT/step_time.txt
timeout -k 10 120 python3 tools/phase_profile.py 4096 > $OUT/phase_profile_lean_n4096.txt 2>&1; cat $OUT/phase_profile_lean_n4096.txt
timeout -k 10 300 python3 bench.py --agents 4096 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_n4096.json 2> $OUT/bench_n4096.err; python3 -c "
import json;d=json.load(open('$OUT/bench_n4096.json'));print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['admm']['us_per_iteration'], d['row_free_step']['ms_per_step'])"
echo "== config 5: nap-only host waits, more workers than cores"
TRIALS=256 EXTRA="--host-wait 2" bash tools/batch_rate.sh $OUT/batch128_hostwait2.txt "4:4 4:6 4:8 5:8" > /dev/null 2>&1; grep "procs\|all . ranks\|errors" $OUT/batch128_hostwait2.txt
echo "== done"
