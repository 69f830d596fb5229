for cfg in "1 3" "3 1" "2 2" "4 1" "3 2" "6 1" "4 2"; do set -- $cfg
python bench.py --model e2evn --batch $1 --streams $2 --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 streams $2', round(d['value'],1), round(d['ms_per_step'],3))
"
done
