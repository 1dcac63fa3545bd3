# Developer experiment (profiling build): how many of the last time steps the linearisation grid cuts in half- / quarter-size
# chunks (SCP_PAIR_TAIL="half quarter"; "0 0" = equal chunks, unset = the host's default rule).  profiles/r03_pair_tail_sweep.txt
# also holds the persistent-grid variant of the same item list (one workgroup per occupancy slot taking items from a
# counter: 139 registers, 3 workgroups per CU), which lost and was removed from the source.
mkdir -p gpurun_out/tail
export SCP_HIP_LIB=$PWD/ba-path-planning_amd/lib/libscp_hip_prof.so
for T in "0 0" "8 4" "4 2" "8 0" "2 1"; do
  echo "== SCP_PAIR_TAIL=$T"
  SCP_PAIR_TAIL="$T" timeout 60 python tools/pair_timeline.py --reps 8 2>&1 | grep "launch\|end times\|resident"
done
