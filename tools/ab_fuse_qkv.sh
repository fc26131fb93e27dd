#!/bin/bash
# A/B of the round-6 encoder fusions on the GPU box (through gpurun, from the repo root):  tools/ab_fuse_qkv.sh [tag] [utterances]
# rocprofv3 --kernel-trace --stats of one offline encoder pass alone (tools/encoder_traffic.py) with
#   0: SIMULST_FUSE_QKV=0 (separate Q | K | V launch)   1: the default (the next layer's Q | K | V rows in the feed-forward launch)
#   2: SIMULST_FUSE_OUT=1, only with docs/patches/r06_out_projection_in_ffn_launch.patch applied (... and the attention output
#      projection in front of it; measured 24.5-24.7 ms against 23.6-23.8 and not kept -- without the patch level 2 repeats level 1)
# writes gpurun_out/<tag>/enc_<B>_fuse<0|1|2>_kernel_stats.csv and prints the totals.   LEVELS="0 1" tools/ab_fuse_qkv.sh ...
TAG=${1:-r06_qkv}; B=${2:-1280}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in ${LEVELS:-0 1 0 1}; do
  unset SIMULST_FUSE_QKV SIMULST_FUSE_OUT
  if [ $V = 0 ]; then export SIMULST_FUSE_QKV=0 SIMULST_FUSE_OUT=0; fi
  if [ $V = 1 ]; then export SIMULST_FUSE_OUT=0; fi
  if [ $V = 2 ]; then export SIMULST_FUSE_OUT=1; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_$V -- python3 $R/tools/encoder_traffic.py $B > $O/enc_$V.log 2>&1; echo "rc=$?"
  S=$(ls $O/enc_$V/*/*kernel_stats.csv | head -1); cp $S $O/enc_${B}_fuse${V}_kernel_stats.csv
  rm -rf $O/enc_$V
  python3 - $O/enc_${B}_fuse${V}_kernel_stats.csv $V <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("fusion level %s: %.2f ms per pass (two passes recorded)" % (sys.argv[2], tot / 2e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("   %-70s calls %4s  avg %8.1f us  total %7.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
