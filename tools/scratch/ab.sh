B="python bench.py --no-cpu-baseline --steps 300 --warmup 100"
for v in base NOPRE LATE_TF NOPREDPEDN_EXP_LATE_TF; do
  if [ $v = base ]; then unset PEDN_HIP_LIB; else export PEDN_HIP_LIB=$PWD/pednstream_amd/csrc/exp_$v.so; fi
  for n in melbourne delft; do echo "== $v $n"; timeout -k 10 120 $B --network $n || exit 1; done
done
