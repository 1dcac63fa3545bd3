#!/bin/bash
# round-3 GPU session d: phase profile of the lean kernel, in-kernel clocks of the linearisation kernel, rho-interval sweep,
# config 5 with the lean kernel
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3d
mkdir -p $OUT
cd $R
echo "== phase profile, lean kernel at 4096 and (forced) at 1024"
timeout -k 10 120 python3 tools/phase_profile.py 4096 > $OUT/phase_profile_lean_n4096.txt 2>&1; cat $OUT/phase_profile_lean_n4096.txt
timeout -k 10 120 python3 tools/phase_profile.py 1024 2 > $OUT/phase_profile_lean_n1024.txt 2>&1; cat $OUT/phase_profile_lean_n1024.txt
echo "== in-kernel clock of the linearisation kernel (profiling build)"
SCP_HIP_LIB=$R/ba-path-planning_amd/lib/libscp_hip_prof.so timeout -k 10 200 python3 tools/pair_context.py --agents 1024 > $OUT/pair_context_clock_1024.txt 2>&1; cat $OUT/pair_context_clock_1024.txt
echo "== rho interval sweep"
timeout -k 10 500 python3 tools/rho_interval_sweep.py > $OUT/rho_interval_sweep.txt 2>&1; tail -12 $OUT/rho_interval_sweep.txt
echo "== config 5 unit: default kernel vs lean"
TRIALS=256 bash tools/batch_rate.sh $OUT/batch128_default.txt "4:4" > /dev/null 2>&1; grep "all 4 ranks\|errors" $OUT/batch128_default.txt
TRIALS=256 EXTRA="--qp-persistent 2" bash tools/batch_rate.sh $OUT/batch128_lean.txt "4:4 4:6 5:6" > /dev/null 2>&1; grep "procs\|all . ranks\|errors" $OUT/batch128_lean.txt
echo "== done"
