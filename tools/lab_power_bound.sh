# Is the step bound by its dependency chain or by the chip's power budget?  An idle gap of known length is put on the forward ->
# backward chain (one spinning wave: no power to speak of); a chain-bound step grows by the gap, a power-bound one by less
# (the clock recovers in the gap and the busy parts run faster).  Alternating runs on one box; prints ms per step, mean / p10 sclk.
python - <<'PY'
import torch, time
torch.cuda.init()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
print('CAL: 1e6 _sleep cycles = %.1f us' % (e0.elapsed_time(e1) * 1e3))
PY
for i in 1 2 3; do
  for US in 0 300 1000; do
    SVOL_BENCH_TURN_SLEEP_US=$US python bench.py --steps 20 --warmup 5 --blocks 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SLEEP_US $US', round(d['ms_per_step'],3), 'sclk', d['sclk_ghz']['mean'], d['sclk_ghz']['p10'], 'attn_bwd_ms', round(d['roofline']['attn_bwd_ms'],4), 'attn_fwd_ms', round(d['roofline']['attn_fwd_ms'],4))" || exit 1
  done
done
