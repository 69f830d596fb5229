for st in 1 2; do for ff in 0 1; do
MRIDC_AMD_FUSED_FINAL=$ff python bench.py --streams $st --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $st fused $ff', round(d['value'],2), round(d['ms_per_step'],3), d['breakdown_ms'])
"
done; done
