#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of one offline encoder pass at <B> utterances (default 1 280) -> gpurun_out/<tag>/encoder_traffic_<B>.json
# (tools/encoder_traffic.py --summarise).  Counter passes carry --kernel-trace only; the program itself follows `--`.  usage: tools/encoder_traffic_pmc.sh [B] [tag]
TAG=${2:-r05_e}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
B=${1:-1280}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/enc_fetch -- python3 $R/tools/encoder_traffic.py $B > $O/enc_fetch.log 2>&1; echo "enc fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/enc_write -- python3 $R/tools/encoder_traffic.py $B > $O/enc_write.log 2>&1; echo "enc write rc=$?"
cd $R
F=$(ls $O/enc_fetch/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/enc_write/*/*counter_collection.csv 2>/dev/null | head -1)
python tools/encoder_traffic.py --summarise "$F" "$W" $B $O/encoder_traffic_$B.json
rm -rf $O/enc_fetch $O/enc_write
