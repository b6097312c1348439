#!/bin/bash
# GPU-vs-oracle fuzz campaigns of a round (run through gpurun from the repo root); log -> gpurun_out/profiles_out/<tag>_gpu_fuzz.txt
TAG=${1:?round tag}
O=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/profiles_out
mkdir -p $O
{
  echo "# tools/gpu_fuzz.py / gpu_fuzz_rl.py on the MI355X: every field, the turning fractions and the error flags of 3 replicas per"
  echo "# random network against the CPU oracle (observations and rewards against the restated RL glue), bit for bit"
  PEDN_FUSE_TP=1 python3 tools/gpu_fuzz.py 9000 9600
  PEDN_FUSE_TP=0 python3 tools/gpu_fuzz.py 9600 9900
  PEDN_FUSE_TP=1 PEDN_TF_GENERAL=3 PEDN_TF_LDS_LIMIT=1 python3 tools/gpu_fuzz.py 9900 10200
  PEDN_FUSE_TP=1 python3 tools/gpu_fuzz.py 10200 10500 scenarios
  echo "# the link update inside node_kernel (PEDN_FUSE_LINK=1, last arriver) and node_kernel unrolled for 8 corridors (PEDN_NODE_MD=8):"
  PEDN_FUSE_LINK=1 python3 tools/gpu_fuzz.py 10700 11000
  PEDN_NODE_MD=8 python3 tools/gpu_fuzz.py 11000 11200
  echo "# the two halves of the batch as two chains of launches on two streams against one chain (256 replicas per network):"
  python3 tools/gpu_fuzz_chains.py 12000 12150
  echo "# assign_flows_type 'optimal' (node LP):"
  PEDN_FUZZ_OPTIMAL=1 python3 tools/gpu_fuzz.py 10500 10700
  python3 tools/gpu_fuzz_rl.py 9000 9400
  PEDN_RL_FOLD=0 python3 tools/gpu_fuzz_rl.py 9400 9600
} 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_gpu_fuzz.txt
