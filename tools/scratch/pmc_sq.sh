set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NET=${1:-delft}
CMD="python3 $R/bench.py --network $NET --no-cpu-baseline --steps 40 --warmup 20"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/sq1_$NET --output-format csv -- $CMD > $R/gpurun_out/sq1_$NET.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d $R/gpurun_out/sq2_$NET --output-format csv -- $CMD > $R/gpurun_out/sq2_$NET.log 2>&1
