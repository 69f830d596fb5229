for a in f16x2 bf16x3; do for u in 14x2 18x4; do MRIDC_AMD_ARITH=$a python bench.py --model e2evn --unet $u --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a $u', round(d['value'],1)); 
for k,v in sorted(d.get('breakdown_ms',{}).items(), key=lambda kv:-kv[1] if isinstance(kv[1],(int,float)) else 0)[:16]: print('   ',k, round(v*1e3,1) if isinstance(v,(int,float)) else v)"; done; done
