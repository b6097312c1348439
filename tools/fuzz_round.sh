#!/bin/bash
# GPU-vs-oracle fuzz campaigns of a round (run through gpurun from the repo root); log -> gpurun_out/profiles_out/<tag>_gpu_fuzz.txt
#   bash tools/fuzz_round.sh r04 13000 [k] [part]   k: every campaign k times as long (default 1)
#                                                   part: a = launch variants of the two-launch plan, b = owner-wave plan, node LP, RL;
#                                                   default both (at k = 10 one part is ~10 minutes: one gpurun call each)
TAG=${1:?round tag}
K=${3:-1}
PART=${4:-ab}
O=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/profiles_out
mkdir -p $O
S=${2:-9000}     # first seed of the campaign (round 2: 9000, round 3: 13000 / 100000 / 200000)
part_a() {
  echo "# tools/gpu_fuzz.py on the MI355X: every field, the turning fractions and the error flags of 3 replicas per random network"
  echo "# against the CPU oracle, bit for bit"
  PEDN_FUSE_TP=1 python3 -u tools/gpu_fuzz.py $S $((S+600*K))
  PEDN_FUSE_TP=0 python3 -u tools/gpu_fuzz.py $((S+600*K)) $((S+900*K))
  PEDN_FUSE_TP=1 PEDN_TF_GENERAL=3 PEDN_TF_LDS_LIMIT=1 python3 -u tools/gpu_fuzz.py $((S+900*K)) $((S+1200*K))
  PEDN_FUSE_TP=1 python3 -u tools/gpu_fuzz.py $((S+1200*K)) $((S+1500*K)) scenarios
  echo "# every turning-fraction workgroup in front of the link update (PEDN_TF_HEAVY_GROUPS=0):"
  PEDN_TF_HEAVY_GROUPS=0 python3 -u tools/gpu_fuzz.py $((S+2500*K)) $((S+2700*K))
  echo "# node_kernel unrolled for 8 corridors (PEDN_NODE_MD=8):"
  PEDN_NODE_MD=8 python3 -u tools/gpu_fuzz.py $((S+2000*K)) $((S+2200*K))
  echo "# the two halves of the batch as two chains of launches on two streams against one chain (256 replicas per network):"
  python3 -u tools/gpu_fuzz_chains.py $((S+3000*K)) $((S+3150*K))
}
part_b() {
  echo "# owner-wave plan forced on for every model (PEDN_LINK_OWNER=1: node_kernel<LU>(t + 1) performs the link update of t),"
  echo "# with stand-alone turning fractions (every step flushes), per-replica scenarios, MD = 8; forced off; two chains == one chain:"
  PEDN_LINK_OWNER=1 python3 -u tools/gpu_fuzz.py $((S+3200*K)) $((S+3600*K))
  PEDN_LINK_OWNER=1 PEDN_FUSE_TP=0 python3 -u tools/gpu_fuzz.py $((S+3800*K)) $((S+3900*K))
  PEDN_LINK_OWNER=1 python3 -u tools/gpu_fuzz.py $((S+3900*K)) $((S+4100*K)) scenarios
  PEDN_LINK_OWNER=1 PEDN_NODE_MD=8 python3 -u tools/gpu_fuzz.py $((S+4100*K)) $((S+4200*K))
  PEDN_LINK_OWNER=0 python3 -u tools/gpu_fuzz.py $((S+4200*K)) $((S+4400*K))
  PEDN_LINK_OWNER=1 python3 -u tools/gpu_fuzz_chains.py $((S+4400*K)) $((S+4500*K))
  echo "# the single-launch plan of small batches with HELPER waves (node_kernel_h: waves 8..15 of a workgroup compute the rows of turning"
  echo "# fractions of slot waves 0..7) is the default for these 3-replica networks wherever their rows allow it, i.e. in every campaign"
  echo "# above; here forced OFF, and with the slot waves computing their own rows (node_kernel<LU, TF>, PEDN_INLINE_TF=1):"
  PEDN_INLINE_TF=0 python3 -u tools/gpu_fuzz.py $((S+4500*K)) $((S+4800*K))
  PEDN_INLINE_TF=1 python3 -u tools/gpu_fuzz.py $((S+4800*K)) $((S+5000*K))
  PEDN_INLINE_TF=1 python3 -u tools/gpu_fuzz.py $((S+5600*K)) $((S+5700*K)) scenarios
  echo "# lazy reset against the ordinary reset under random sequences of calls (tools/gpu_fuzz_lazy.py):"
  python3 -u tools/gpu_fuzz_lazy.py $((S+5000*K)) $((S+5200*K))
  echo "# the engine's own launch plans against two launches per step on one stream under random sequences of calls (tools/gpu_fuzz_plans.py):"
  python3 -u tools/gpu_fuzz_plans.py $((S+5200*K)) $((S+5600*K))
  echo "# assign_flows_type 'optimal' (node LP):"
  PEDN_FUZZ_OPTIMAL=1 python3 -u tools/gpu_fuzz.py $((S+1500*K)) $((S+1700*K))
  echo "# tools/gpu_fuzz_rl.py: observations and rewards of the batched RL step against the restated RL glue:"
  python3 -u tools/gpu_fuzz_rl.py $S $((S+400*K))
  PEDN_RL_FOLD=0 python3 -u tools/gpu_fuzz_rl.py $((S+400*K)) $((S+600*K))
  echo "# ... with every step but the first through the device-resident step clock (the launches a captured graph replays):"
  python3 -u tools/gpu_fuzz_rl.py $((S+600*K)) $((S+1000*K)) clocked
  PEDN_RL_FOLD=0 python3 -u tools/gpu_fuzz_rl.py $((S+1000*K)) $((S+1200*K)) clocked
}
# line-buffered: a block-buffered grep held everything back until the script ended, the run looked silent for more than 7 minutes and
# was killed (gpurun_out/fuzz_d_call.log of round 3); the campaigns also print a heartbeat line every 100-200 seeds
{
  case $PART in *a*) part_a ;; esac
  case $PART in *b*) part_b ;; esac
} 2>&1 | grep --line-buffered -v amdgpu.ids | tee -a $O/${TAG}_gpu_fuzz.txt
