mkdir -p gpurun_out/r3n/pmc; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $R/gpurun_out/r3n/pmc/sq_names.txt
export SCP_HIP_LIB=$R/ba-path-planning_amd/lib/libscp_hip_prof.so
for AB in 0 1; do
  export SCP_PAIR_ABLATE=$AB
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/r3n/pmc/a${AB}_p1 -- python3 $R/tools/pair_bench.py --reps 3 > $R/gpurun_out/r3n/pmc/a${AB}_p1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/r3n/pmc/a${AB}_p2 -- python3 $R/tools/pair_bench.py --reps 3 > $R/gpurun_out/r3n/pmc/a${AB}_p2.log 2>&1
done
find $R/gpurun_out/r3n/pmc -name "*counter_collection.csv" | head
