R=$PWD; O=$R/gpurun_out/r05_e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1280 448; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_$B -- python3 $R/tools/encoder_traffic.py $B > $O/enc_$B.log 2>&1; echo "rc=$?"
S=$(ls $O/enc_$B/*/*kernel_stats.csv | head -1); cp $S $O/enc_${B}_kernel_stats.csv
rm -rf $O/enc_$B
done
