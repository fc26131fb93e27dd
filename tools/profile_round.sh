#!/bin/bash
# One measurement point on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh <tag>
# Writes under gpurun_out/<tag>/ : the driver-shaped bench line, the default bench line, rocprofv3 kernel-trace stats of
# the driver-shaped command, three PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) of a 4096-row sequence, the scan
# kernels under rocprofv3.  The program itself follows `--` (no wrapper), --pmc only with --kernel-trace.
TAG=${1:-r02_a}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench k20 rc=$?"
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_k20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20_under_rocprofv3.json 2> $O/trace_k20.err; echo "trace rc=$?"
PMCCMD="$R/bench.py --steps 64 --warmup 64 --concurrency 1 --no-pipeline --no-cpu-baseline --timed-only --min-warmup-seconds 0"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $PMCCMD > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $PMCCMD > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $PMCCMD > $O/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scan -- python3 $R/tools/scan_bench.py > $O/scan_bench.json 2> $O/scan.err; echo "scan rc=$?"
cd $R
F=$(ls $O/pmc_fetch/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv 2>/dev/null | head -1); M=$(ls $O/pmc_mfma/*/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_classes.py traffic $F $W 4096 $O/pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 64 --warmup 64 --concurrency 1 --no-pipeline --no-cpu-baseline --timed-only --min-warmup-seconds 0" > $O/pmc_traffic.txt
[ -n "$M" ] && python tools/pmc_classes.py mfma $M $O/pmc_mfma.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 64 --warmup 64 --concurrency 1 --no-pipeline --no-cpu-baseline --timed-only --min-warmup-seconds 0" > $O/pmc_mfma.txt
# keep the merged directory small: the stats csv files and the JSON summaries travel, the raw counter dumps do not
for d in pmc_fetch pmc_write pmc_mfma; do rm -rf $O/$d; done
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -size +8M -delete
ls -la $O $O/trace_k20/* $O/scan/* 2>/dev/null | head -40
