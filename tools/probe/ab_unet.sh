python tools/probe/unet_conv_time.py 2>&1 | tail -10
for f in 0 1; do for u in 14x2 18x4; do
MRIDC_AMD_UNET_FUSED=$f python bench.py --model e2evn --unet $u --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused $f unet $u', round(d['value'],1), round(d['ms_per_step'],3))
"
done; done
