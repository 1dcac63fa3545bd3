# developer experiment: tail classes of the linearisation grid (profiling builds: one item per workgroup / persistent grid)
mkdir -p gpurun_out/r3n
for L in prof profp; do
export SCP_HIP_LIB=$PWD/ba-path-planning_amd/lib/libscp_hip_$L.so
for T in "0 0" "8 4" "4 2" "8 0" "2 1"; do
  echo "== $L SCP_PAIR_TAIL=$T"
  SCP_PAIR_TAIL="$T" timeout 60 python tools/pair_timeline.py --reps 8 2>&1 | grep "launch\|end times\|resident"
done
done
