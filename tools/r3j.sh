#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3j
mkdir -p $OUT
cd $R
echo "== persistent-kernel tests (all variants)"
timeout -k 10 600 python -m pytest tests/test_qp_gpu.py tests/test_scp_gpu.py -m gpu -q -k "persistent or rho_switch or lean or config4 or 3d_beyond" > $OUT/tests.log 2>&1; tail -8 $OUT/tests.log
echo "== step times: 8-agent kernel vs the lean state diet with 8 agents per workgroup"
timeout -k 10 300 python3 tools/step_time.py 128x2 128x2x3 1024x2 1024x2x3 2048x2 2048x2x3 512x3 512x3x3 1024x3 1024x3x3 1100x3 2048x3 > $OUT/step_time.txt 2>&1; cat $OUT/step_time.txt
echo "== phase profile of the lean 8-agent form at 1024"
timeout -k 10 120 python3 tools/phase_profile.py 1024 3 > $OUT/phase_profile_lean8_n1024.txt 2>&1; cat $OUT/phase_profile_lean8_n1024.txt
echo "== done"
