# A/B of one environment switch on the default bench, alternating runs on ONE box:  bash tools/ab_bench.sh VAR=VALUE [pairs] [extra bench args]
# prints the median-of-blocks ms per step of every run (A = switch unset, B = switch set)
SW=$1; PAIRS=${2:-3}; shift; shift
for i in $(seq $PAIRS); do
  python bench.py --steps 20 --warmup 5 --blocks 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A', d['ms_per_step'], d['ms_per_step_blocks'])" || exit 1
  env $SW python bench.py --steps 20 --warmup 5 --blocks 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', d['ms_per_step'], d['ms_per_step_blocks'])" || exit 1
done
