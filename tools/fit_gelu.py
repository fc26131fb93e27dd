#!/usr/bin/env python3
"""Coefficients of the bf16 paths' GELU (csrc/common.h gelu_fast2, csrc/ffn_pipe.hip gelu_q1..q4), CPU only.

    GELU(x) = h + |h| - |h| erfc(|h| sqrt 2),  h = x / 2,   erfc(|h| sqrt 2) ~= exp2(Q(|h|)),  Q(a) = c1 a + ... + c5 a^5

Q is a weighted minimax fit (Lawson iterations over weighted least squares) of log2 erfc on a = |h| in (0, 9]; the weight is what an
error of Q does to the RESULT, d GELU = |h| erfc ln2 dQ, so the fit spends its accuracy where the result is sensitive and none where
erfc has vanished.  Prints the fp32 coefficients, the maximum |error of the result| in exact arithmetic and in a simulated fp32
evaluation (Horner with fused multiply-adds, exp2 in double rounded to fp32: v_exp_f32 is good to 1 ulp), the same for the
Abramowitz-Stegun 7.1.28 form the kernels used through round 5, and checks that Q falls monotonically (c5 < 0: exp2 underflows to 0
for any magnitude, no clamp needed).
"""
import numpy as np
from scipy.special import erf, erfc


def gelu(x):
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def fit(deg=5, A=9.0, iters=200, n=400001):
    a = np.linspace(1e-7, A, n)
    f = np.log2(erfc(a * np.sqrt(2)).clip(1e-300))
    w = a * erfc(a * np.sqrt(2)) * np.log(2)
    V = np.vander(a, deg + 1, increasing=True)[:, 1:]            # no constant term: Q(0) = 0, erfc(0) = 1 exactly
    lw, best = np.ones_like(a), None
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(V * (w * lw)[:, None], f * w * lw, rcond=None)
        e = np.abs((V @ c - f) * w)
        if best is None or e.max() < best[0]:
            best = (e.max(), c.copy())
        lw = lw * (e / e.max() + 1e-3) ** 0.5
        lw /= lw.max()
    return best


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def main():
    err, c = fit()
    c32 = c.astype(np.float32)
    print("max |error of the result|, exact arithmetic:", err)
    for i, v in enumerate(c32, 1):
        print(f"  c{i} = {float(v)!r}f   ({hex(int(v.view(np.uint32)))})")
    x = np.linspace(-20, 20, 4000001).astype(np.float32)
    h = (x * np.float32(0.5)).astype(np.float32)
    a = np.abs(h)
    k = [np.full_like(a, v) for v in c32]
    q = fma(a, k[4], k[3])
    for j in (2, 1, 0):
        q = fma(q, a, k[j])
    e = np.exp2((q * a).astype(np.float32).astype(np.float64)).astype(np.float32)
    g = fma(-a, e, (h + a).astype(np.float32))
    ref = gelu(x.astype(np.float64))
    d = np.abs(g.astype(np.float64) - ref)
    print("fp32 evaluation: max |error|", d.max(), "at x =", float(x[d.argmax()]))
    z = a.astype(np.float64) * np.sqrt(2)
    p = 1 + z * (0.0705230784 + z * (0.0422820123 + z * (0.0092705272 + z * (0.0001520143 + z * (0.0002765672 + z * 0.0000430638)))))
    print("Abramowitz-Stegun 7.1.28 form (rounds 2-5): max |error|", np.abs(h.astype(np.float64) + a - a / p ** 16 - ref).max())
    t = np.linspace(0, 40, 100001)
    assert c[-1] < 0 and np.all(np.diff(np.vander(t, 6, increasing=True)[:, 1:] @ c) < 0), "Q must fall monotonically"
    print("Q monotone on [0, 40], leading coefficient < 0: exp2(Q) -> 0 for large |x|")


if __name__ == "__main__":
    main()
