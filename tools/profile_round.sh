#!/bin/bash
# Regenerates the tracked measurement files of a round on the GPU box (run through gpurun from the repo root):
#
#   gpurun --timeout 1150 -- 'bash tools/profile_round.sh r02 [part]'      part: all (default) | trace | bench | extras | rl
#
# writes gpurun_out/profiles_out/<tag>_*; copy those into profiles/ afterwards.  Kernel trace and every PMC group are
# separate rocprofv3 runs of the same bench command (a --pmc run must not be combined with other trace domains).
set -e
TAG=${1:?round tag, e.g. r02}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
P=$O/profiles_out
mkdir -p $P
cd /tmp
export TMPDIR=/tmp

profile() {  # $1 = suffix ("" or "_delft"), rest = bench arguments
  local SUF=$1; shift
  # counter collection serialises kernels: without this the stream-overlap probe falls back to one chain and the passes would count
  # whole-batch launches instead of the default plan's half-batch ones
  export PEDN_STREAM_PROBE=0
  rm -rf $O/kt$SUF $O/pf$SUF $O/pw$SUF $O/sq1$SUF $O/sq2$SUF $O/sq3$SUF $O/sq4$SUF
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$SUF -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $O/kt$SUF.log 2>&1
  echo "kernel trace$SUF done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf$SUF -- python3 $R/bench.py "$@" --steps 48 --warmup 20 --no-cpu-baseline --no-extra > $O/pf$SUF.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw$SUF -- python3 $R/bench.py "$@" --steps 48 --warmup 20 --no-cpu-baseline --no-extra > $O/pw$SUF.log 2>&1
  echo "FETCH/WRITE$SUF done"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq1$SUF -- python3 $R/bench.py "$@" --steps 48 --warmup 20 --no-cpu-baseline --no-extra > $O/sq1$SUF.log 2>&1 || echo "SQ pass 1 failed (see sq1$SUF.log)"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/sq2$SUF -- python3 $R/bench.py "$@" --steps 48 --warmup 20 --no-cpu-baseline --no-extra > $O/sq2$SUF.log 2>&1 || echo "SQ pass 2 failed (see sq2$SUF.log)"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq3$SUF -- python3 $R/bench.py "$@" --steps 48 --warmup 20 --no-cpu-baseline --no-extra > $O/sq3$SUF.log 2>&1 || echo "SQ pass 3 failed (see sq3$SUF.log)"
  rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq4$SUF -- python3 $R/bench.py "$@" --steps 48 --warmup 20 --no-cpu-baseline --no-extra > $O/sq4$SUF.log 2>&1 || echo "SQ pass 4 failed (see sq4$SUF.log)"
  unset PEDN_STREAM_PROBE
  (cd $R && python3 tools/summarize_busy.py $O/sq3$SUF $O/sq4$SUF > $P/$TAG${SUF}_unit_busy.json) || true
  echo "SQ$SUF done"
  (cd $R && python3 tools/summarize_profiles.py $TAG$SUF $O/kt$SUF $O/pf$SUF $O/pw$SUF $O/cf $O/cw 20 $O/kt$SUF.log)
  (cd $R && python3 tools/summarize_sq.py $O/sq1$SUF $O/sq2$SUF --skip 20 > $P/$TAG${SUF}_sq_counters.json) || true
  cp $R/profiles/$TAG${SUF}_kernel_stats.csv $R/profiles/$TAG${SUF}_pmc.json $P/
}

if [ "$PART" = all ] || [ "$PART" = trace ]; then
  rm -rf $O/cf $O/cw $O/sc
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cf -- python3 $R/tools/pmc_calibrate.py > $O/cf.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cw -- python3 $R/tools/pmc_calibrate.py > $O/cw.log 2>&1
  profile "" --network melbourne
  # the same command with ONE chain of launches (PEDN_STREAMS=1): every launch covers the whole batch and runs alone -- the per-launch
  # figure a reader can check by hand (bytes of the whole batch / mean launch duration)
  rm -rf $O/kt1
  PEDN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python3 $R/bench.py --network melbourne --no-cpu-baseline --no-extra --no-live-traffic > $O/kt1.log 2>&1
  cp $O/kt1/*/*_kernel_stats.csv $P/${TAG}_one_chain_kernel_stats.csv
  profile _delft --network delft
  # config #5: per-kernel durations of the batched RL step, plain and after reset(options={'randomize': True})
  rm -rf $O/ktrl $O/ktrl2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrl -- python3 $R/bench.py --rl --network 45_intersections --replicas 2048 > $O/ktrl.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrl2 -- python3 $R/bench.py --rl --randomize --network 45_intersections --replicas 2048 > $O/ktrl2.log 2>&1
  cp $O/ktrl/*/*_kernel_stats.csv $P/${TAG}_rl_kernel_stats.csv
  cp $O/ktrl2/*/*_kernel_stats.csv $P/${TAG}_rl_randomized_kernel_stats.csv
  rm -rf $O/ktrl $O/ktrl2
  # practical ceiling of a coalesced stream (8- / 16-byte lanes, scattered chunks): per-launch durations of device_math_kernel
  rocprofv3 --kernel-trace --output-format csv -d $O/sc -- python3 $R/tools/stream_ceiling.py > $O/sc.log 2>&1
  python3 - "$O/sc" > $P/${TAG}_stream_ceiling.txt <<'PY'
import csv, glob, sys
rows = [r for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0])) if "device_math_kernel" in r["Kernel_Name"]]
labels = ["8-byte lanes", "16-byte lanes"] * 3 + [f"scattered chunks of {c * 8} B" for c in (32, 64, 128, 256, 512, 4096)]
print("a[i] + b[i] -> out[i] over 2^26 doubles (1 GiB read + 0.5 GiB written per launch), tools/stream_ceiling.py under rocprofv3 --kernel-trace")
for r, lab in zip(rows, labels):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"  {lab:32s} {us:9.1f} us  {1.5 * 2**30 / us / 1e6:7.2f} TB/s")
PY
  rm -rf $O/kt* $O/pf* $O/pw* $O/sq1* $O/sq2* $O/sq3* $O/sq4* $O/cf $O/cw $O/sc      # the raw traces are large; the summaries above are what is kept
fi
if [ "$PART" = rl ]; then      # only the RL kernel statistics of the trace part
  rm -rf $O/ktrl $O/ktrl2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrl -- python3 $R/bench.py --rl --network 45_intersections --replicas 2048 > $O/ktrl.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrl2 -- python3 $R/bench.py --rl --randomize --network 45_intersections --replicas 2048 > $O/ktrl2.log 2>&1
  cp $O/ktrl/*/*_kernel_stats.csv $P/${TAG}_rl_kernel_stats.csv
  cp $O/ktrl2/*/*_kernel_stats.csv $P/${TAG}_rl_randomized_kernel_stats.csv
  rm -rf $O/ktrl $O/ktrl2
  head -4 $P/${TAG}_rl_kernel_stats.csv $P/${TAG}_rl_randomized_kernel_stats.csv | cut -c1-150
fi
cd $R
if [ "$PART" = all ] || [ "$PART" = bench ]; then
  # the default command: its LAST stdout line is the compact object the driver parses, everything measured goes to bench_full.json
  python3 bench.py > $P/${TAG}_bench_n1_line.json 2> $O/bench.err
  cp bench_full.json $P/${TAG}_bench_n1.json
  python3 bench.py --steps 20 --warmup 5 > $P/${TAG}_bench_driver_command_line.json 2>> $O/bench.err        # the driver's own command
  python3 bench.py --network delft > $P/${TAG}_delft_bench_n1_line.json 2>> $O/bench.err
  cp bench_full.json $P/${TAG}_delft_bench_n1.json
  python3 bench.py --rl --network 45_intersections --replicas 2048 > $P/${TAG}_bench_rl_config5.json 2>> $O/bench.err
  python3 bench.py --rl --network 45_intersections --replicas 2048 --history recent >> $P/${TAG}_bench_rl_config5.json 2>> $O/bench.err
  python3 bench.py --rl --randomize --network 45_intersections --replicas 2048 >> $P/${TAG}_bench_rl_config5.json 2>> $O/bench.err
  for r in 2048 3072 4096; do python3 bench.py --replicas $r --no-cpu-baseline --no-extra --steps 200 --warmup 50; done > $P/${TAG}_replica_scaling.jsonl 2>> $O/bench.err
  { python3 bench.py --network nine_intersections --replicas 256 --no-cpu-baseline --no-extra; python3 bench.py --network 45_intersections --replicas 2048 --no-cpu-baseline --no-extra; } > $P/${TAG}_small_configs.jsonl 2>> $O/bench.err
fi
if [ "$PART" = all ] || [ "$PART" = extras ]; then
  make -s -C pednstream_amd/csrc phase-profile 2>> $O/bench.err || true      # the instrumented build must match the engine's ABI
  python3 tools/phase_profile.py melbourne delft 45_intersections:2048 nine_intersections:256 > $P/${TAG}_phase_profile.txt 2>> $O/bench.err
  PEDN_FUSE_TP=0 python3 tools/turn_phase_profile.py delft >> $P/${TAG}_phase_profile.txt 2>> $O/bench.err
  python3 tools/turn_phase_profile.py delft | sed 's/^== /== (inside link_turn_kernel) /' >> $P/${TAG}_phase_profile.txt 2>> $O/bench.err
  python3 tools/dropin_time.py > $P/${TAG}_dropin_time.txt 2>> $O/bench.err
  python3 tools/reset_time.py > $P/${TAG}_reset_time.txt 2>> $O/bench.err
  python3 tools/reset_time.py 45_intersections 2048 recent >> $P/${TAG}_reset_time.txt 2>> $O/bench.err
  /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/xstream tools/xstream_bench.hip 2> /dev/null && timeout -k 5 60 /tmp/xstream > $P/${TAG}_cross_stream_dependency.txt 2>&1 || true
  /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rates tools/valu_rates.hip 2> /dev/null && timeout -k 5 60 /tmp/valu_rates > $P/${TAG}_valu_rates.txt 2>&1 || true
  # stream ceilings by footprint / mix / access width, the per-group barrier, the workgroup timeline of link_turn_kernel
  /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/mall_stream tools/mall_stream.hip 2> /dev/null && timeout -k 5 200 /tmp/mall_stream > $P/${TAG}_mall_stream_raw.txt 2>&1 || true
  /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/group_barrier tools/group_barrier.hip 2> /dev/null && timeout -k 5 120 /tmp/group_barrier > $P/${TAG}_group_barrier_raw.txt 2>&1 || true
  timeout -k 5 200 python3 tools/lt_timeline.py delft 1024 > $P/${TAG}_lt_timeline_raw.txt 2>> $O/bench.err || true
  python3 -m pytest tests -q -m gpu > $P/${TAG}_pytest_gpu.log 2>&1
fi
ls -la $P
