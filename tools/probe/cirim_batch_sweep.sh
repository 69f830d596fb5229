for cfg in "1 2" "2 1" "2 2" "4 1" "4 2" "8 1"; do set -- $cfg
python bench.py --batch $1 --streams $2 --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 streams $2', round(d['value'],2), round(d['ms_per_step'],3), d['breakdown_ms'])
"
done
