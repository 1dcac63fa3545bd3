#!/bin/bash
# Developer tool: everything that backs the numbers in DESIGN.md / profiles/README.md, in one run on the GPU box.
#   gpurun -- 'bash tools/collect_profiles.sh r03'      -> gpurun_out/<tag>/...   (copy the summaries into profiles/)
# rocprofv3 runs from /tmp (its scratch files), counters in their own passes (--kernel-trace only next to --pmc).
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$R/ba-path-planning_amd
cd /tmp

echo "== bench (with the CPU baseline)"
timeout -k 10 400 python3 $R/bench.py --steps 20 --warmup 2 > $OUT/bench_n1024.json 2> $OUT/bench_n1024.err
tail -c 600 $OUT/bench_n1024.json; echo

echo "== kernel stats of the bench (the JSON line printed under rocprof carries the HIP-event duration of the same launches)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 10 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
cp $(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1) $OUT/bench_n1024_kernel_stats.csv 2>/dev/null
grep '^{"metric"' $OUT/prof_bench.log | tail -1 > $OUT/bench_n1024_under_rocprof.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench4096 -- python3 $R/bench.py --agents 4096 --steps 5 --no-cpu-baseline > $OUT/prof_bench4096.log 2>&1
cp $(find $OUT/prof_bench4096 -name "*kernel_stats.csv" | head -1) $OUT/bench_n4096_1gpu_kernel_stats.csv 2>/dev/null

echo "== HBM counters of the pairwise passes (separate passes)"
for C in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $R/tools/pair_bench.py --reps 3 > $OUT/pmc_$C.log 2>&1
  cp $(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1) $OUT/pairwise_pmc_$C.csv 2>/dev/null
done
for C in WRITE_SIZE FETCH_SIZE; do  # config 4 (4096 x 50: the L1 / L2 path of the same kernels)
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${C}_n4096 -- python3 $R/tools/pair_bench.py --reps 3 --agents 4096 > $OUT/pmc_${C}_n4096.log 2>&1
  cp $(find $OUT/pmc_${C}_n4096 -name "*counter_collection.csv" | head -1) $OUT/pairwise_pmc_${C}_n4096.csv 2>/dev/null
done
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_TCC -- python3 $R/tools/pair_bench.py --reps 3 > $OUT/pmc_TCC.log 2>&1
cp $(find $OUT/pmc_TCC -name "*counter_collection.csv" | head -1) $OUT/pairwise_pmc_TCC.csv 2>/dev/null

echo "== matrix-core counters of the QP kernels"
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counter_names.txt
cat $OUT/mfma_counter_names.txt | tr '\n' ' '; echo
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfma.log 2>&1
cp $(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/qp_pmc_mfma_raw.csv 2>/dev/null
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_lds.log 2>&1
cp $(find $OUT/pmc_lds -name "*counter_collection.csv" | head -1) $OUT/qp_pmc_lds_raw.csv 2>/dev/null

cd $R
echo "== phase profiles of the persistent kernels, step times, box calibration"
timeout -k 10 200 python3 tools/phase_profile.py 1024 > $OUT/phase_profile_n1024.txt 2>&1
timeout -k 10 200 python3 tools/phase_profile.py 1024 4 > $OUT/phase_profile_round2_kernel_n1024.txt 2>&1
head -12 $OUT/phase_profile_n1024.txt
timeout -k 10 200 python3 tools/phase_profile.py 4096 > $OUT/phase_profile_lean_n4096.txt 2>&1
head -12 $OUT/phase_profile_lean_n4096.txt
timeout -k 10 300 python3 tools/step_time.py 128x2 1024x2 1024x2x4 2048x2 2048x2x4 2056x2 3000x2 4096x2 64x3 512x3 1024x3 1100x3 2048x3 > $OUT/step_time.txt 2>&1; cat $OUT/step_time.txt
timeout -k 10 60 tools/bin/membw > $OUT/membw.txt 2>&1
SCP_HIP_LIB=$R/ba-path-planning_amd/lib/libscp_hip_prof.so timeout -k 10 200 python3 tools/pair_context.py --agents 1024 > $OUT/pair_context_clock_1024.txt 2>&1; cat $OUT/pair_context_clock_1024.txt
echo "== linearisation kernel: workgroup timeline (equal chunks vs the large -> small item list), without its stores, SQ counters"
for T in "0 0" ""; do
  F=$OUT/pair_timeline_$( [ -z "$T" ] && echo items || echo equal_chunks ).txt
  if [ -z "$T" ]; then SCP_HIP_LIB=$R/ba-path-planning_amd/lib/libscp_hip_prof.so timeout -k 10 100 python3 tools/pair_timeline.py --reps 8 > $F 2>&1
  else SCP_PAIR_TAIL="$T" SCP_HIP_LIB=$R/ba-path-planning_amd/lib/libscp_hip_prof.so timeout -k 10 100 python3 tools/pair_timeline.py --reps 8 > $F 2>&1; fi
done
SCP_PAIR_ABLATE=1 SCP_HIP_LIB=$R/ba-path-planning_amd/lib/libscp_hip_prof.so timeout -k 10 100 python3 tools/pair_timeline.py --reps 4 > $OUT/pair_timeline_items_nostores.txt 2>&1
grep "launch [4-7]" $OUT/pair_timeline_equal_chunks.txt $OUT/pair_timeline_items.txt | cut -c1-160
timeout -k 10 200 python3 tools/pair_bench.py --reps 5 --agents 4096 > $OUT/pair_bench_n4096.txt 2>&1
timeout -k 10 200 python3 tools/pair_bench.py --reps 5 --agents 1024 > $OUT/pair_bench_n1024.txt 2>&1
echo "== 2-rank rehearsals on this one GPU (gloo, host-staged exchanges)"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --share-gpu --backend gloo --steps 10 --warmup 2 --no-cpu-baseline 2> $OUT/rehearsal_1024.err | grep '^{"metric"' > $OUT/rehearsal_2ranks_gloo_n1024.json
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 bench.py --agents 4096 --gpus 2 --share-gpu --backend gloo --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/rehearsal_4096.err | grep '^{"metric"' > $OUT/rehearsal_2ranks_gloo_n4096.json
python3 -c "
import json
for f in ('$OUT/rehearsal_2ranks_gloo_n1024.json', '$OUT/rehearsal_2ranks_gloo_n4096.json'):
    d = json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('exchange'))"
echo "== 4096 x 50 on one GPU, full solves, soak, batch"
timeout -k 10 300 python3 bench.py --agents 4096 --steps 5 --warmup 1 > $OUT/bench_n4096_1gpu.json 2> $OUT/bench_n4096.err
timeout -k 10 200 python3 tools/full_solve_timing.py 64 256 1024 4096 > $OUT/full_solve_timing.txt 2>&1
timeout -k 10 200 python3 tools/ref_config_timing.py > $OUT/ref_config_timing.txt 2>&1
timeout -k 10 300 python3 tools/soak.py 80 > $OUT/soak_80.txt 2>&1; tail -1 $OUT/soak_80.txt
timeout -k 10 300 python3 tools/soak.py 80 --polish > $OUT/soak_80_polish.txt 2>&1; tail -1 $OUT/soak_80_polish.txt
timeout -k 10 400 python3 tools/soak.py 24 --large > $OUT/soak_24_large.txt 2>&1; tail -1 $OUT/soak_24_large.txt
# config 5's unit: 128-agent scenarios per second on this ONE GPU, processes x streams, steady state (--warmup 1), and the
# cold figure (solver creation and kernel loading inside the clock) for one process
TRIALS=256 bash tools/batch_rate.sh $OUT/batch128_rates.txt "1:1 1:4 2:4 4:4 4:5" > /dev/null 2>&1
TRIALS=96 WARMUP=0 bash tools/batch_rate.sh $OUT/batch128_cold.txt "1:1 1:4" > /dev/null 2>&1
# the same layouts with the kernel-timing events on and with the multi-launch form of the pairwise passes (what the one-launch
# passes and the missing events are worth), and what the records of the 4 x 5 run say (pipelines, give-ups, stragglers)
python3 tools/batch_records_summary.py /tmp/b_4_5 4 > $OUT/batch128_records_4x5.txt 2>&1
TRIALS=256 EXTRA="--kernel-timing 1" bash tools/batch_rate.sh $OUT/batch128_with_events.txt "1:4 4:5" > /dev/null 2>&1
SCP_NO_SMALL_PASS=1 TRIALS=256 bash tools/batch_rate.sh $OUT/batch128_multi_launch_passes.txt "1:1 1:4 4:5" > /dev/null 2>&1
cat $OUT/batch128_rates.txt $OUT/batch128_cold.txt
cd /tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_solve128 -- python3 $R/tools/solve128.py 128 8 > $OUT/solve128.log 2>&1
cp $(find $OUT/prof_solve128 -name "*kernel_stats.csv" | head -1) $OUT/solve128_kernel_stats.csv 2>/dev/null
grep "^solve" $OUT/solve128.log > $OUT/solve128.txt
cp $(find $OUT/prof_solve128 -name "*kernel_trace.csv" | head -1) $OUT/solve128_kernel_trace.csv 2>/dev/null
cd $R
python3 tools/solve_timeline.py $OUT/solve128_kernel_trace.csv > $OUT/solve128_timeline.txt 2>&1
SCP_HIP_LIB=$R/ba-path-planning_amd/lib/libscp_hip_prof.so timeout -k 10 100 python3 tools/small_pass_profile.py 128 > $OUT/small_pass_profile_n128.txt 2>&1
SCP_HIP_LIB=$R/ba-path-planning_amd/lib/libscp_hip_prof.so timeout -k 10 100 python3 tools/small_pass_profile.py 256 > $OUT/small_pass_profile_n256.txt 2>&1
cat $OUT/small_pass_profile_n128.txt
timeout -k 10 100 python3 tools/demo_k500.py > $OUT/demo_k500.txt 2>&1; tail -5 $OUT/demo_k500.txt
timeout -k 10 60 tools/bin/grid_sync_bench 2000 > $OUT/grid_sync_bench.txt 2>&1
echo "== done"
