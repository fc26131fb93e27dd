#!/bin/bash
O=gpurun_out/r04_af
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "rc=$?"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r04_af/bench_k20.json").read().strip().splitlines()[-1])
c4=j['configs4_rank_shard']
print(j['value'], c4['offline']['tokens_per_s'], c4['offline'].get('passes_s_this_rank'), c4['streaming_evaluation']['tokens_per_s'], c4['streaming_evaluation'].get('passes_s_this_rank'))
PY
grep -h "passes of\|total" $O/bench_k20.err | tail -3 | cut -c1-200
