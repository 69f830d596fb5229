for cfg in "4 2" "6 2" "8 2" "4 3" "8 1"; do set -- $cfg
python bench.py --model e2evn --batch $1 --streams $2 --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 streams $2', round(d['value'],1))
"
done
