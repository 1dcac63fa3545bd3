#!/bin/bash
# Developer tool: config 5's unit (128-agent grid-swap scenarios) per second on ONE GPU, for several process x stream
# combinations (processes via torch.distributed.run, all on the same card; at most 5 workers + the launcher use the GPU).
# usage: [EXTRA="--qp-persistent 2"] tools/batch_rate.sh OUTFILE "P:S P:S ..."      e.g. "1:1 1:4 2:4 4:4"
OUT=${1:-/tmp/batch_rate.txt}
COMBOS=${2:-"1:1 1:4 2:4 4:4"}
TRIALS=${TRIALS:-96}
export PYTHONPATH=$PWD/ba-path-planning_amd:$PYTHONPATH
: > $OUT
for PS in $COMBOS; do
  P=${PS%%:*}; S=${PS##*:}
  echo "procs $P streams $S" >> $OUT
  if [ "$P" = "1" ]; then
    timeout -k 10 250 python3 -m path_planning.cli.compute_trajectories_batch --Ns 128 --trials $TRIALS --scenario grid-swap --seed 1 \
      --results-dir /tmp/b_${P}_${S} --streams $S --warmup ${WARMUP:-1} ${EXTRA:-} 2>&1 | grep "scenarios/s\|errors=" >> $OUT
  else
    timeout -k 10 250 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 \
      --master-port $((29600 + P * 10 + S)) -m path_planning.cli.compute_trajectories_batch --Ns 128 --trials $((TRIALS * P)) \
      --scenario grid-swap --seed 1 --results-dir /tmp/b_${P}_${S} --streams $S --warmup ${WARMUP:-1} ${EXTRA:-} 2>&1 | grep "scenarios/s\|errors=" >> $OUT
  fi
done
cat $OUT
