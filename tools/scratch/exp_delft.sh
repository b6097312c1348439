set -e
B="python bench.py --network delft --no-cpu-baseline --steps 200 --warmup 50"
for v in base NODYN NOTFWRITE; do
  if [ $v = base ]; then unset PEDN_HIP_LIB; else export PEDN_HIP_LIB=$PWD/pednstream_amd/csrc/exp_$v.so; fi
  echo "== $v"; timeout -k 10 120 $B
done
unset PEDN_HIP_LIB
echo "== meanfield"; timeout -k 10 120 $B --rng-mode meanfield
