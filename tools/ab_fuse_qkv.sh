#!/bin/bash
# A/B of the next layer's Q | K | V projection inside the feed-forward launch (SIMULST_FUSE_QKV) on the GPU box (through gpurun, from the
# repo root):  tools/ab_fuse_qkv.sh [tag] [utterances]   -- rocprofv3 --kernel-trace --stats of one offline encoder pass alone
# (tools/encoder_traffic.py) with the switch off and on; writes gpurun_out/<tag>/enc_<B>_qkv<0|1>_kernel_stats.csv and prints the totals.
TAG=${1:-r06_qkv}; B=${2:-1280}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in 0 1 0 1; do
  export SIMULST_FUSE_QKV=$V
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_$V -- python3 $R/tools/encoder_traffic.py $B > $O/enc_$V.log 2>&1; echo "rc=$?"
  S=$(ls $O/enc_$V/*/*kernel_stats.csv | head -1); cp $S $O/enc_${B}_qkv${V}_kernel_stats.csv
  rm -rf $O/enc_$V
  python3 - $O/enc_${B}_qkv${V}_kernel_stats.csv $V <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("FUSE_QKV=%s: %.2f ms per pass (two passes recorded)" % (sys.argv[2], tot / 2e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:9]:
    print("   %-70s calls %4s  avg %8.1f us  total %7.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
