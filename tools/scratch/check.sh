set -e
timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --steps 300 --warmup 100"
echo "== melbourne"; timeout -k 10 120 $B
echo "== delft"; timeout -k 10 120 $B --network delft
echo "== 45_intersections"; timeout -k 10 120 $B --network 45_intersections --replicas 2048
