#!/bin/bash
O=gpurun_out/r04_ad
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_encoder.py -x -q -m gpu 2>&1 | tail -3
bash tools/_exp.sh 2>&1 | grep -v "amdgpu.ids"
for v in 1 0 1 0; do
  SIMULST_PANEL_WIDE=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/b_$v.json 2> $O/b_$v.err
  echo "panel_wide $v: $(grep -h 'passes of' $O/b_$v.err | cut -c1-160)"
done
for v in 1 0; do
  SIMULST_PANEL_WIDE=$v timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/d_$v.json 2> $O/d_$v.err
  echo "default run, panel_wide $v: $(grep -h 'passes of' $O/d_$v.err | cut -c1-160)"
done
