#!/bin/bash
# Regenerates the tracked measurement files of a round on the GPU box (run through gpurun from the repo root):
#
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r01'
#
# writes gpurun_out/profiles_out/<tag>_*; copy those into profiles/ afterwards.  Kernel trace and the two PMC counters are
# separate rocprofv3 runs of the same bench command (a --pmc run must not be combined with other trace domains).
set -e
TAG=${1:?round tag, e.g. r01}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
P=$O/profiles_out
mkdir -p $P
cd /tmp
export TMPDIR=/tmp
rm -rf $O/kt* $O/pf* $O/pw* $O/cf $O/cw

profile() {  # $1 = suffix ("" or "_delft"), rest = bench arguments
  local SUF=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$SUF -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/kt$SUF.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf$SUF -- python3 $R/bench.py "$@" --steps 48 --warmup 20 --no-cpu-baseline > $O/pf$SUF.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw$SUF -- python3 $R/bench.py "$@" --steps 48 --warmup 20 --no-cpu-baseline > $O/pw$SUF.log 2>&1
  (cd $R && python3 tools/summarize_profiles.py $TAG$SUF $O/kt$SUF $O/pf$SUF $O/pw$SUF $O/cf $O/cw 20)
  cp $R/profiles/$TAG${SUF}_kernel_stats.csv $R/profiles/$TAG${SUF}_pmc.json $P/
}

rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cf -- python3 $R/tools/pmc_calibrate.py > $O/cf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cw -- python3 $R/tools/pmc_calibrate.py > $O/cw.log 2>&1
profile "" --network melbourne
profile _delft --network delft
cd $R
python3 bench.py > $P/${TAG}_bench_n1.json 2> $O/bench.err                    # the driver's command (reads the fresh PMC summary)
python3 bench.py --network delft --no-cpu-baseline > $P/${TAG}_delft_bench_n1.json 2>> $O/bench.err
python3 bench.py --rl --network 45_intersections --replicas 2048 > $P/${TAG}_bench_rl_config5.json 2>> $O/bench.err
python3 tools/phase_profile.py melbourne delft > $P/${TAG}_phase_profile.txt 2>> $O/bench.err
python3 -m pytest tests -q -m gpu > $P/${TAG}_pytest_gpu.log 2>&1
rm -rf $O/kt* $O/pf* $O/pw* $O/cf $O/cw      # the raw traces are large; the summaries above are what is kept
ls -la $P
