#!/bin/bash
# Round-6 measurement point on the GPU box (through gpurun, from the repo root):  tools/profile_r06.sh <tag>
# Writes under gpurun_out/<tag>/: the driver-shaped bench line + bench_legs.json, rocprofv3 --kernel-trace --stats of the driver-shaped
# command (main leg only), FETCH_SIZE / WRITE_SIZE passes of one encoder pass (tools/encoder_traffic.py, 256 utterances) and of the
# dominant kernel (tools/kernel_bench.py cross_attn), a SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE pass of the main leg.
# Counter passes carry --kernel-trace only; the program itself follows `--` (no wrapper).
TAG=${1:-r06}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench k20 rc=$?"; cp bench_legs.json $O/bench_k20_legs.json
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; cp bench_legs.json $O/bench_default_legs.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_k20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_k20_under_rocprofv3.json 2> $O/trace_k20.err; echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/enc_fetch -- python3 $R/tools/encoder_traffic.py 256 > $O/enc_fetch.log 2>&1; echo "enc fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/enc_write -- python3 $R/tools/encoder_traffic.py 256 > $O/enc_write.log 2>&1; echo "enc write rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/enc_fetch_1280 -- python3 $R/tools/encoder_traffic.py 1280 > $O/enc_fetch_1280.log 2>&1; echo "enc fetch 1280 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/enc_write_1280 -- python3 $R/tools/encoder_traffic.py 1280 > $O/enc_write_1280.log 2>&1; echo "enc write 1280 rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 20 --warmup 5 --concurrency 1 --no-pipeline --no-cpu-baseline --no-extra-configs --timed-only --min-warmup-seconds 0 > $O/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/kernel_bench.py cross_attn --utterances 448 4096 > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/kernel_bench.py cross_attn --utterances 448 4096 > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R
timeout 120 python tools/kernel_bench.py cross_attn --utterances 448 4096 > $O/kernel_bench_cross_attn.json 2>/dev/null
F=$(ls $O/enc_fetch/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/enc_write/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/encoder_traffic.py --summarise "$F" "$W" 256 $O/encoder_traffic.json; fi
F=$(ls $O/enc_fetch_1280/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/enc_write_1280/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/encoder_traffic.py --summarise "$F" "$W" 1280 $O/encoder_traffic_1280_utterances.json; fi
F=$(ls $O/pmc_fetch/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/pmc_cross_attn.py "$F" "$W" $O/kernel_bench_cross_attn.json $O/pmc_cross_attention_traffic.json; fi
M=$(ls $O/pmc_mfma/*/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$M" ] && python tools/pmc_kernel.py $O/pmc_mfma.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 20 --warmup 5 --concurrency 1 --no-pipeline --no-cpu-baseline --no-extra-configs --timed-only --min-warmup-seconds 0" "." "$M" > /dev/null
S=$(ls $O/trace_k20/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp "$S" $O/k20_kernel_stats.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/enc_fetch $O/enc_write $O/enc_fetch_1280 $O/enc_write_1280
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
grep -h "passes of" $O/*.err | cut -c1-160
ls $O
