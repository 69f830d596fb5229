#!/bin/bash
O=gpurun_out/r04k; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt | cut -c1-250
timeout 900 python tools/probe/train_parity.py > $O/parity_headline.txt 2>&1; grep -E "seed|whole|oracle vs" $O/parity_headline.txt | cut -c1-250
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04k/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], "streamed", (d.get("streamed_inputs") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "parity", (d.get("parity_vs_oracle") or {}).get("rel_l2"))
for k, v in (d.get("other_configs") or {}).items():
    print("  ", k, v.get("value"), (v.get("parity_vs_oracle") or {}).get("rel_l2"), v.get("error"))
t = d["other_configs"]["cirim_training_bf16_15coil_640x372"]
print(json.dumps(t.get("parity_vs_oracle"))[:1500])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_train -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py $O/prof_train/t_results.db | head -16 | cut -c1-150
