#!/bin/bash
# SQ counters of the fused feed-forward launch at 1 280 utterances (through gpurun, from the repo root): tools/ffn_pmc.sh [tag]
# -> gpurun_out/<tag>/pmc_ffn.json.  Counter passes carry --kernel-trace only; the program itself follows `--`.
TAG=${1:-r06_ffn}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -- python3 $R/tools/ffn_once.py > $O/p$i.log 2>&1; echo "pass $i ($P) rc=$?"
done
cd $R
python tools/pmc_kernel.py $O/pmc_ffn.json "rocprofv3 --kernel-trace --pmc <4 counters per pass> -- python3 tools/ffn_once.py (1 280 utterances x 378 rows)" "ffn_pipe_kernel" $(ls $O/p*/*/*counter_collection.csv) | cut -c1-2500
rm -rf $O/p[0-9]*/
