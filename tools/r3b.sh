#!/bin/bash
# round-3 GPU session b: tests, bench with the row-free step beside it, 2-rank rehearsals, box calibration
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3b
mkdir -p $OUT
cd $R
echo "== gpu tests"; timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; tail -5 $OUT/gputest.log
echo "== box calibration"; (rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -8) > $OUT/clocks.txt 2>&1; cat $OUT/clocks.txt
timeout -k 10 60 tools/bin/membw > $OUT/membw.txt 2>&1; grep -E "wr_linear|wr_chunk2d16 |rd_linear" $OUT/membw.txt | tail -6
timeout -k 10 120 python3 tools/pair_context.py --agents 1024 > $OUT/pair_context_1024.txt 2>&1; cat $OUT/pair_context_1024.txt
echo "== bench 1024"; timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $OUT/bench_n1024.json 2> $OUT/bench_n1024.err; python3 - <<PY
import json
d=json.load(open("$OUT/bench_n1024.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["row_free_step"], d["config"]["qp"], d.get("parity_max_abs"))
PY
echo "== 2 ranks on one GPU (gloo), 1024"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --share-gpu --backend gloo --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_n1024_2ranks_gloo.json 2> $OUT/bench_n1024_2ranks.err; tail -c 400 $OUT/bench_n1024_2ranks.err; python3 -c "
import json;d=json.loads(open('$OUT/bench_n1024_2ranks_gloo.json').read().strip().splitlines()[-1]);print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['config']['qp'])"
echo "== 4096: 1 rank, then 2 ranks on one GPU (gloo)"
timeout -k 10 300 python3 bench.py --agents 4096 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_n4096.json 2> $OUT/bench_n4096.err; python3 -c "
import json;d=json.load(open('$OUT/bench_n4096.json'));print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['row_free_step'], d['config']['qp'])"
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 bench.py --agents 4096 --gpus 2 --share-gpu --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_n4096_2ranks_gloo.json 2> $OUT/bench_n4096_2ranks.err; tail -c 400 $OUT/bench_n4096_2ranks.err; python3 -c "
import json;d=json.loads(open('$OUT/bench_n4096_2ranks_gloo.json').read().strip().splitlines()[-1]);print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['config']['qp'])"
echo "== done"
