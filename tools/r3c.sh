#!/bin/bash
# round-3 GPU session c: the lean persistent kernel (small tests first), the in-kernel clock of the linearisation kernel
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3c
mkdir -p $OUT
cd $R
echo "== lean kernel: QP-level tests"
timeout -k 10 300 python -m pytest tests/test_qp_gpu.py -m gpu -x -q -k "persistent or rho_switch" > $OUT/lean_qp.log 2>&1; rc=$?; tail -15 $OUT/lean_qp.log
if [ $rc -ne 0 ]; then echo "lean QP tests failed (rc $rc): stopping"; exit 1; fi
echo "== lean kernel: solves"
timeout -k 10 400 python -m pytest tests/test_scp_gpu.py -m gpu -q -k "lean or config4 or carry" > $OUT/lean_scp.log 2>&1; rc=$?; tail -15 $OUT/lean_scp.log
if [ $rc -ne 0 ]; then echo "lean solve tests failed (rc $rc): going on"; fi
echo "== step times"
timeout -k 10 200 python3 tools/step_time.py 1024x2 2048x2 2056x2 3000x2 4096x2 > $OUT/step_time.txt 2>&1; cat $OUT/step_time.txt
echo "== bench 4096"
timeout -k 10 300 python3 bench.py --agents 4096 --steps 5 --warmup 1 > $OUT/bench_n4096.json 2> $OUT/bench_n4096.err; python3 -c "
import json;d=json.load(open('$OUT/bench_n4096.json'));print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['admm'], d['row_free_step'], d['config']['qp'], d.get('parity_max_abs'))"
echo "== in-kernel clock of the linearisation kernel (profiling build)"
SCP_HIP_LIB=$R/ba-path-planning_amd/lib/libscp_hip_prof.so timeout -k 10 200 python3 tools/pair_context.py --agents 1024 > $OUT/pair_context_clock_1024.txt 2>&1; cat $OUT/pair_context_clock_1024.txt
echo "== ADMM step experiments"
timeout -k 10 300 python3 tools/admm_steps_exp.py --agents 1024 > $OUT/admm_steps_exp_1024.txt 2>&1; cat $OUT/admm_steps_exp_1024.txt
echo "== whole gpu suite"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; tail -5 $OUT/gputest.log
echo "== done"
