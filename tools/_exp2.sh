#!/bin/bash
O=gpurun_out/r04_ae
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "panel or emformer or encoder" 2>&1 | tail -3
python - <<'PY'
import torch, time, sys, statistics
sys.path.insert(0, '.')
from simulst_amd.ops import Ops
from simulst_amd import _lib
from simulst_amd._lib import EPI_BIAS, EPI_BIAS_RES
ops = Ops()
g = torch.Generator().manual_seed(1)
bf = torch.bfloat16
def once(f, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
ops.h.set_option(_lib.OPT_PANEL_WIDE, 0)
for M, N, epi in ((441600, 768, EPI_BIAS), (422400, 256, EPI_BIAS_RES), (110000, 768, EPI_BIAS)):
    x = (torch.randn(M, 256, generator=g) * 0.5).to(bf).cuda()
    W = (torch.randn(N, 256, generator=g) / 16).to(bf).cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).to(bf).cuda() if epi == EPI_BIAS_RES else None
    Wp = ops.pack_fragment_major(W)
    y = torch.empty(M, N, dtype=bf, device="cuda")
    f = lambda: ops.linear(x, Wp, b, epilogue=epi, residual=R, out=y, w_fragment_major=True)
    for _ in range(20): f()
    ts = [once(f) for _ in range(12)]
    print(M, N, epi, "panel (32 rows/wave) %.1f us" % statistics.median(ts), flush=True)
PY
for v in 1 1; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/b_$v.json 2> $O/b_$v.err
  echo "bench: $(grep -h 'passes of' $O/b_$v.err | cut -c1-160)"
done
