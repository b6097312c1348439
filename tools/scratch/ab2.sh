V=$1
export PEDN_HIP_LIB=$PWD/pednstream_amd/csrc/exp_$V.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 || exit 1
B="python bench.py --no-cpu-baseline --steps 300 --warmup 100"
for v in base $V; do
  if [ $v = base ]; then unset PEDN_HIP_LIB; else export PEDN_HIP_LIB=$PWD/pednstream_amd/csrc/exp_$v.so; fi
  for n in melbourne delft; do echo "== $v $n"; timeout -k 10 120 $B --network $n || exit 1; done
done
