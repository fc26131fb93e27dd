TAG=r02_e
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench k20 rc=$?"
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python tools/eval_sharded.py --utterances 5000 > $O/config5_shard.json 2> $O/config5.err; echo "config5 rc=$?"; tail -1 $O/config5_shard.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_k20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20_under_rocprofv3.json 2> $O/trace_k20.err; echo "trace rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 64 --warmup 64 --concurrency 1 --no-pipeline --no-cpu-baseline --timed-only --min-warmup-seconds 0 > $O/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
cd $R
M=$(ls $O/pmc_mfma/*/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$M" ] && python tools/pmc_classes.py mfma $M $O/pmc_mfma.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 64 --warmup 64 --concurrency 1 --no-pipeline --no-cpu-baseline --timed-only --min-warmup-seconds 0" > $O/pmc_mfma.txt
rm -rf $O/pmc_mfma; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
grep "timed region" $O/*.err; head -4 $O/pmc_mfma.txt
python tools/kernel_bench.py emf_attn > $O/kernel_bench_emf_attn.json 2>/dev/null; python tools/kernel_bench.py self_attn > $O/kernel_bench_self_attn.json 2>/dev/null; python tools/determinism_check.py > $O/determinism_check.json 2>/dev/null; python tools/determinism_check.py --rows 192 --steps 60 >> $O/determinism_check.json 2>/dev/null
cat $O/determinism_check.json | cut -c1-120
