#!/bin/bash
# GPU-vs-oracle fuzz campaigns of a round (run through gpurun from the repo root); log -> gpurun_out/profiles_out/<tag>_gpu_fuzz.txt
#   bash tools/fuzz_round.sh r03 13000 [k]        k: every campaign k times as long (default 1)
TAG=${1:?round tag}
K=${3:-1}
O=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/profiles_out
mkdir -p $O
{
  echo "# tools/gpu_fuzz.py / gpu_fuzz_rl.py on the MI355X: every field, the turning fractions and the error flags of 3 replicas per"
  echo "# random network against the CPU oracle (observations and rewards against the restated RL glue), bit for bit"
  S=${2:-9000}     # first seed of the campaign (round 2: 9000, round 3: 13000)
  PEDN_FUSE_TP=1 python3 tools/gpu_fuzz.py $S $((S+600*K))
  PEDN_FUSE_TP=0 python3 tools/gpu_fuzz.py $((S+600*K)) $((S+900*K))
  PEDN_FUSE_TP=1 PEDN_TF_GENERAL=3 PEDN_TF_LDS_LIMIT=1 python3 tools/gpu_fuzz.py $((S+900*K)) $((S+1200*K))
  PEDN_FUSE_TP=1 python3 tools/gpu_fuzz.py $((S+1200*K)) $((S+1500*K)) scenarios
  echo "# stand-alone link update with two replicas per lane in one / two segments (PEDN_LINK_NS=1|2; the default is one replica per lane),"
  echo "# every turning-fraction workgroup in front of the link update (PEDN_TF_HEAVY_GROUPS=0):"
  PEDN_LINK_NS=1 python3 tools/gpu_fuzz.py $((S+2200*K)) $((S+2400*K))
  PEDN_LINK_NS=2 python3 tools/gpu_fuzz.py $((S+2400*K)) $((S+2500*K))
  PEDN_TF_HEAVY_GROUPS=0 python3 tools/gpu_fuzz.py $((S+2500*K)) $((S+2700*K))
  echo "# node_kernel unrolled for 8 corridors (PEDN_NODE_MD=8):"
  PEDN_NODE_MD=8 python3 tools/gpu_fuzz.py $((S+2000*K)) $((S+2200*K))
  echo "# the two halves of the batch as two chains of launches on two streams against one chain (256 replicas per network):"
  python3 tools/gpu_fuzz_chains.py $((S+3000*K)) $((S+3150*K))
  echo "# assign_flows_type 'optimal' (node LP):"
  PEDN_FUZZ_OPTIMAL=1 python3 tools/gpu_fuzz.py $((S+1500*K)) $((S+1700*K))
  python3 tools/gpu_fuzz_rl.py $S $((S+400*K))
  PEDN_RL_FOLD=0 python3 tools/gpu_fuzz_rl.py $((S+400*K)) $((S+600*K))
} 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_gpu_fuzz.txt
