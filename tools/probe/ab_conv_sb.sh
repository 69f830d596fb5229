python tools/probe/conv_sb_time.py 2>&1 | tail -4
for m in ccnn vsnet rvn; do for f in 0 1; do
MRIDC_AMD_ARITH=$([ $f = 1 ] && echo f16x2 || echo fp32) python bench.py --model $m --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m conv_sb $f', round(d['value'],1))
"
done; done
