for t4 in 0 1; do for s in 1 2; do
MRIDC_AMD_LLG_T4=$t4 python bench.py --mask 2d --no-cpu-baseline --steps 10 --warmup 3 --streams $s 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('t4=$t4 streams=$s', round(d['value'],2), {k:round(v*1e3,1) for k,v in d['breakdown_ms'].items() if k!='rim_steps_per_slice'})"
done; done
