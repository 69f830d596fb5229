for st in 1 2; do for ip in 0 1; do
MRIDC_AMD_INPLACE_STATE=$ip python bench.py --streams $st --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $st inplace $ip', round(d['value'],2), round(d['ms_per_step'],3), d['breakdown_ms'])
"
done; done
