for st in 2 3 4; do
python bench.py --streams $st --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $st', round(d['value'],2), round(d['ms_per_step'],3))
"
done
