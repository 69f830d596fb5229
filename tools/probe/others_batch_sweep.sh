for m in qcirim rvn ccnn vsnet; do for cfg in "1 2" "2 2" "4 2" "4 1"; do set -- $cfg
python bench.py --model $m --batch $1 --streams $2 --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m batch $1 streams $2', round(d['value'],1))
"
done; done
