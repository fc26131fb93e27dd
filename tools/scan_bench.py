#!/usr/bin/env python3
"""The north_star-named scan kernels at the configs[1] / configs[3] shapes, for rocprofv3 (program directly after `--`):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d <out> -- python3 /root/repo/tools/scan_bench.py

  expected_alignment_kernel   (1536, 110, 32)   fp32  -- 64 utterances x 4 heads x 6 layers, 110 targets, 32 pooled keys
                              (1536, 110, 250)        -- the same rows without pre-decision pooling
  soft_attention / mass_preservation on the same rows
  step_search_kernel          (4096 x 4 rows, 32) and (.., 250): one decode step of a 4096-row launch sequence
  cif_kernel                  [64, 250] and [64, 1500] alpha ~ U(0,1) (SURVEY 8(d) config 4 stress input), C = 256, bf16 + fp32;
                              [1024, 250] the co-scheduled shape
Each kernel is launched REPS times after a warm-up launch; tensors are allocated once, outside the launches.  Prints one
JSON line with the algorithmic bytes per launch (8 B / element for the alignment scans, 4BS + 2BSC read +
2BT'C written for CIF, 4 B / element + 16 B / row for the step search) so that the rocprofv3 average durations convert
to GB/s against the 8 TB/s HBM peak.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

REPS = 20


def main():
    from simulst_amd.ops import Ops
    ops = Ops()
    g = torch.Generator().manual_seed(999)
    out = {}
    for S in (32, 250):
        p = torch.sigmoid(torch.randn(1536, 110, S, generator=g) * 2).cuda()
        e = (torch.randn(1536, 110, S, generator=g) * 3).cuda()
        for _ in range(REPS + 1):
            alpha = ops.expected_alignment(p, None, 1e-6)
        for _ in range(REPS + 1):
            amp = ops.mass_preservation(alpha.clone(), None)
        for _ in range(REPS + 1):
            ops.expected_soft_attention(amp, e, None, None, 1e-10)
        n = p.numel()
        out[f"expected_alignment_1536x110x{S}"] = {"bytes": 8 * n}
        out[f"mass_preservation_1536x110x{S}"] = {"bytes": 8 * n}
        out[f"soft_attention_1536x110x{S}"] = {"bytes": 12 * n}
        rows = 4096 * 4
        ps = torch.sigmoid(torch.randn(rows, S, generator=g) * 2).cuda()
        hs = torch.zeros(rows, dtype=torch.int64, device="cuda")
        sl = torch.full((rows,), S, dtype=torch.int32, device="cuda")
        for _ in range(REPS + 1):
            hs.zero_()
            ops.mma_step_search(ps, hs, src_len=sl, mass_preservation=True, want_alpha=False)
        out[f"step_search_{rows}x{S}"] = {"bytes": 4 * rows * S + 16 * rows}
    for (B, S), dt in (((64, 250), torch.bfloat16), ((64, 1500), torch.bfloat16), ((64, 1500), torch.float32),
                       ((1024, 250), torch.bfloat16)):
        x = torch.randn(B, S, 256, generator=g).to(dt).cuda()
        a = torch.rand(B, S, generator=g).cuda()
        for _ in range(REPS + 1):
            res = ops.cif_integrate(x, a, beta=1.0, tail_thres=0.5)
        n_out = int(res[1].max())
        esz = x.element_size()
        out[f"cif_{B}x{S}_{'bf16' if dt == torch.bfloat16 else 'f32'}"] = {
            "bytes": 4 * B * S + esz * B * S * 256 + esz * B * n_out * 256}
    torch.cuda.synchronize()
    print(json.dumps({"reps": REPS, "hbm_peak_GBps": 8000, "launches": out}))


if __name__ == "__main__":
    main()
