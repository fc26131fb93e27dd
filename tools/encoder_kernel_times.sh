#!/bin/bash
# Per-kernel times of ONE offline encoder pass alone on the GPU (through gpurun, from the repo root):  tools/encoder_kernel_times.sh [tag]
# rocprofv3 --kernel-trace --stats of tools/encoder_traffic.py at 1 280 and 448 utterances (two passes each; the first allocates);
# writes gpurun_out/<tag>/enc_<B>_kernel_stats.csv and prints the top kernels (tools/kernel_stats_summary.py).
TAG=${1:-r05_e}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1280 448; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_$B -- python3 $R/tools/encoder_traffic.py $B > $O/enc_$B.log 2>&1; echo "rc=$?"
  S=$(ls $O/enc_$B/*/*kernel_stats.csv | head -1); cp $S $O/enc_${B}_kernel_stats.csv
  rm -rf $O/enc_$B
done
cd $R
python3 tools/kernel_stats_summary.py $TAG _kernel | head -24
