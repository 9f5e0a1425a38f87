# A/B of one bench.py ARGUMENT on the default bench, alternating runs on ONE box:  bash tools/ab_bench_arg.sh --some-flag [pairs]
# prints the median-of-blocks ms per step of every run (A = without the flag, B = with it)
FLAG=$1; PAIRS=${2:-3}
for i in $(seq $PAIRS); do
  python bench.py --steps 20 --warmup 5 --blocks 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A', d['ms_per_step'], d['ms_per_step_blocks'])" || exit 1
  python bench.py --steps 20 --warmup 5 --blocks 5 --no-cpu-baseline $FLAG 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', d['ms_per_step'], d['ms_per_step_blocks'])" || exit 1
done
