run() { MRIDC_AMD_ARITH=$([ $1 = 1 ] && echo f16x2 || echo bf16x3) python bench.py --no-cpu-baseline --steps 12 --warmup 4 $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('layer1_f16=$1 $2', round(d['value'],2), {k:round(v*1e3,1) for k,v in d['breakdown_ms'].items() if k!='rim_steps_per_slice' and v})"; }
run 1; run 1; run 1 "--streams 1"; run 0; run 0; run 0 "--streams 1"
