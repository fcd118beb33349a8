#!/usr/bin/env python3
"""Numerics of a split-f16 product whose two CROSS terms are carried in fp8 (e4m3, MX-style power-of-two scale per 32-element block along k) instead of f16
-- the arithmetic `v_mfma_scale_f32_16x16x128_f8f6f4` would run at twice the f16 rate -- next to the shipped three-term form and to dropping the terms
(DESIGN.md section 8 "next"; profiles/r06_exp_dw_cross_terms.txt is the dropped-term measurement on the GPU).  CPU only (numpy); python tools/fp8_cross_numerics.py
Error measure: |result - exact| / sum_k |a_k w_k| per output, exact in float64."""
import numpy as np

rng = np.random.default_rng(0)


def pow2_scale(mx):
    """power of two that lifts the largest |element| into the top f16 binade (the kernels' row scale)"""
    e = np.floor(np.log2(np.maximum(mx, 1e-300)))
    return 2.0 ** (14 - e)


def split_f16(x):
    sc = pow2_scale(np.abs(x).max(axis=1, keepdims=True))
    xs = x * sc
    h1 = xs.astype(np.float16).astype(np.float64)
    h2 = (xs - h1).astype(np.float16).astype(np.float64)
    return h1, h2, sc


def e4m3(x):
    """round to OCP fp8 e4m3 (3 mantissa bits, normal exponents -6..8, subnormals, max 448)"""
    ax = np.abs(x)
    e = np.floor(np.log2(np.maximum(ax, 2.0 ** -20)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)
    return np.sign(x) * np.minimum(np.round(ax / step) * step, 448.0)


def mx_fp8(p):
    """fp8 copy of a plane with one power-of-two scale per 32 elements along k (the MX block of the scaled MFMA)"""
    out = np.empty_like(p)
    for k0 in range(0, p.shape[1], 32):
        blk = p[:, k0:k0 + 32]
        mx = np.abs(blk).max(axis=1, keepdims=True)
        s = 2.0 ** np.floor(np.log2(448.0 / np.maximum(mx, 1e-300)))
        out[:, k0:k0 + 32] = e4m3(blk * s) / s
    return out


def forms(a, w):
    """a [M, K], w [N, K] -> dict of a w^T in each arithmetic (products exact, as the matrix pipe forms them)"""
    a1, a2, sa = split_f16(a)
    w1, w2, sw = split_f16(w)
    un = 1.0 / (sa * sw.T)
    main = a1 @ w1.T
    res = {"three f16 terms (shipped)": (main + a1 @ w2.T + a2 @ w1.T) * un,
           "f16 main + fp8 cross terms": (main + mx_fp8(a1) @ mx_fp8(w2).T + mx_fp8(a2) @ mx_fp8(w1).T) * un,
           "f16 main + one f16 cross term": (main + a1 @ w2.T) * un,
           "f16 main only": main * un}
    return res


def report(title, a, w):
    exact = a @ w.T
    den = np.abs(a) @ np.abs(w).T
    print(title)
    for name, r in forms(a, w).items():
        e = np.abs(r - exact) / den
        print("   %-32s max %.2e   rms %.2e" % (name, e.max(), np.sqrt((e * e).mean())))


# a layer product: 256-long contraction, softplus-like activations against N(0, 2/K) weights
K = 256
a = np.log1p(np.exp(rng.normal(size=(512, K)))) * 0.3
w = rng.normal(size=(256, K)) * np.sqrt(2.0 / K)
report("layer product, K = 256 (512 points x 256 output columns)", a, w)
# a cotangent product: heavy-tailed rows (a few samples carry most of a ray's gradient)
g = rng.normal(size=(512, K)) * np.exp(rng.normal(size=(512, 1)) * 2.0) * 1e-4
report("cotangent x W, K = 256 (rows of very different size)", g, w)
# a weight gradient: contraction over the POINTS of a batch (columns = points here), 2048 (a 16-ray fixture) and 131072 (C3)
for npts in (2048, 131072):
    s = (np.log1p(np.exp(rng.normal(size=(npts, 32)))) * 0.3).T          # [32 S columns, points]
    e = (rng.normal(size=(npts, 32)) * np.exp(rng.normal(size=(npts, 1)) * 2.0) * 1e-4).T
    # the kernels scale per POINT row (32-point tiles): emulate with per-column scales folded in by transposing tiles -- here one scale per operand row
    # of the transposed problem is the optimistic case for every form alike
    report("weight gradient, contraction over %d points (32 x 32 block)" % npts, s, e)
