#!/bin/bash
# Round-3 measurement point on the GPU box (through gpurun, from the repo root):  tools/profile_r03.sh <tag>
# Writes under gpurun_out/<tag>/ (r03_c adds the self-paced streaming runs and the configs[4] shard in both semantics): the driver-shaped bench line (with the configs[2] / configs[3] legs), the default bench line,
# rocprofv3 --kernel-trace --stats of the driver-shaped command (main leg only) and of the batched streaming runs of configs[2] /
# configs[3], two --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) of the dominant kernel.
# The program itself follows `--` (no wrapper).
TAG=${1:-r03_b}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench k20 rc=$?"
timeout 900 python bench.py --no-extra-configs > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_k20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_k20_under_rocprofv3.json 2> $O/trace_k20.err; echo "trace rc=$?"
for c in 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_stream_c$c -- python3 $R/tools/profile_streaming.py --config $c --rows 448 --repeats 2 > $O/stream_c$c.log 2>&1; echo "stream c$c rc=$?"
done
# the evaluation form (self-paced rows, encoder states of one offline forward) of configs[1] / [2] / [3]
for c in 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_paced_c$c -- python3 $R/tools/profile_streaming.py --config $c --rows 448 --repeats 2 --self-paced --encoder offline > $O/paced_c$c.log 2>&1; echo "paced c$c rc=$?"
done
# configs[4]: one rank's shard of the 40 000-utterance set, offline decode and streaming evaluation (wait-k, MMA-hard)
( cd $R; timeout 600 python tools/eval_sharded.py --utterances 5000 2>/dev/null | tail -n 1 > $O/config5_shard_offline.json
  timeout 600 python tools/eval_sharded.py --utterances 5000 --streaming 2>/dev/null | tail -n 1 > $O/config5_shard_streaming_waitk.json
  timeout 600 python tools/eval_sharded.py --utterances 5000 --streaming --policy hard 2>/dev/null | tail -n 1 > $O/config5_shard_streaming_hard.json ); echo "shard rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/kernel_bench.py cross_attn --utterances 448 4096 > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/kernel_bench.py cross_attn --utterances 448 4096 > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R
timeout 120 python tools/kernel_bench.py cross_attn --utterances 448 4096 > $O/kernel_bench_cross_attn.json 2>/dev/null
F=$(ls $O/pmc_fetch/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/pmc_cross_attn.py "$F" "$W" $O/kernel_bench_cross_attn.json $O/pmc_cross_attention_traffic.json; fi
rm -rf $O/pmc_fetch $O/pmc_write
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
grep -h "passes of" $O/*.err | cut -c1-160
ls $O $O/trace_k20/* 2>/dev/null | head -30
