#!/bin/bash
# L2 hit rate and EA (fabric / HBM side) read requests of the decode-step kernels, alone and in the three-stream interleaving
# (through gpurun, from the repo root):  tools/chain_l2_pmc.sh [tag]  ->  gpurun_out/<tag>/chain_l2_in_situ.json
# Counter passes carry --kernel-trace only; the program itself follows `--`.
TAG=${1:-r06_l2}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -o "TCC_[A-Za-z0-9_]*" $O/counters_available.txt | sort -u > $O/tcc_counters.txt
pick() { for c in "$@"; do if grep -qx "$c" $O/tcc_counters.txt; then echo -n "$c "; fi; done; }
P1=$(pick TCC_HIT_sum TCC_MISS_sum)
P2=$(pick TCC_EA0_RDREQ_sum TCC_EA_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA_RDREQ_32B_sum)
P3=$(pick TCC_EA0_RDREQ_DRAM_sum TCC_EA_RDREQ_DRAM_sum TCC_REQ_sum)
echo "passes: [$P1] [$P2] [$P3]"
for S in 1 3; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1)); [ -z "$P" ] && continue
    timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/s${S}_p$i -- python3 $R/tools/decode_l2_probe.py --streams $S > $O/s${S}_p$i.log 2>&1; echo "streams $S pass $i rc=$?"
  done
done
cd $R
A=$(ls $O/s1_p*/*/*counter_collection.csv 2>/dev/null | tr '\n' ','); B=$(ls $O/s3_p*/*/*counter_collection.csv 2>/dev/null | tr '\n' ',')
python tools/decode_l2_probe.py --summarise $O/chain_l2_in_situ.json alone=$A three_streams=$B
rm -rf $O/s1_p* $O/s3_p*/ ; ls $O
