#!/bin/bash
# L2 (TCC) side of the step kernels: hit rate, requests to the fabric and their mean latency (outstanding-request level / requests),
# stalls on fabric credits.  Separate --pmc passes of the bench command, two counters each (more "exceeds the capabilities of the
# hardware to collect"), every pass under its own timeout; a pass that fails or times out stops the script.
#   gpurun -- 'bash tools/l2_counters.sh r03 [network]'   -> gpurun_out/profiles_out/<tag>[_network]_l2_counters.json
TAG=${1:?round tag}
NET=${2:-melbourne}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
P=$O/profiles_out
mkdir -p $P
cd /tmp
export TMPDIR=/tmp
B="python3 $R/bench.py --network $NET --steps 48 --warmup 20 --no-cpu-baseline --no-extra --no-live-traffic"
DIRS=""
i=0
for pair in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
            "TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
            "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  rm -rf $O/l2_$i
  timeout -k 10 150 rocprofv3 --pmc $pair --kernel-trace --output-format csv -d $O/l2_$i -- $B > $O/l2_$i.log 2>&1
  rc=$?
  echo "pass $i ($pair): rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 $O/l2_$i.log; echo "stopping"; break; fi
  DIRS="$DIRS $O/l2_$i"
done
SUF=""; [ "$NET" != melbourne ] && SUF="_$NET"
[ -n "$DIRS" ] && (cd $R && python3 tools/summarize_sq.py $DIRS --skip 20 > $P/${TAG}${SUF}_l2_counters.json) && cat $P/${TAG}${SUF}_l2_counters.json
for d in $DIRS; do rm -rf $d; done
