"""The numerical argument behind the "f32x3" / "f32h2" entry points (include/crnerf.h), restated in numpy so that it is pinned
without a GPU: an fp32 operand split into three bf16 pieces (six leading piece products) or two fp16 pieces (three leading piece products, weights
scaled by 2^8) reproduces the fp32 product to about one fp32 rounding, i.e. the split error sits BELOW the accumulation noise of an fp32 matrix
product.  Piece products are formed exactly (float64) and summed exactly, so what is measured is the representation error of the splits alone --
the part the kernels add on top of fp32 accumulation.  The GPU tests (tests/test_gpu_x3.py, tests/test_gpu_h2.py) measure the kernels themselves
against float64."""
import numpy as np
import torch


def bf16_pieces(x):
    t = torch.from_numpy(x)
    p1 = t.bfloat16().float()
    p2 = (t - p1).bfloat16().float()
    p3 = (t - p1 - p2).bfloat16().float()
    return [p.numpy().astype(np.float64) for p in (p1, p2, p3)]


def fp16_pieces(x, scale=1.0):
    xs = (x * np.float32(scale)).astype(np.float32)
    with np.errstate(over="ignore"):      # out-of-range values become inf, as in the pack kernel
        h1 = xs.astype(np.float16)
    with np.errstate(invalid="ignore"):
        h2 = (xs - h1.astype(np.float32)).astype(np.float16)
    return [h1.astype(np.float64), h2.astype(np.float64)]


def _operands(seed, act_scales=(1e-4, 1e-2, 1.0, 30.0)):
    rng = np.random.default_rng(seed)
    K, M, N = 256, 256, 2048
    W = (rng.standard_normal((M, K)) * 0.1).astype(np.float32)
    X = np.maximum(rng.standard_normal((K, N)), 0).astype(np.float32) * rng.choice(np.float32(act_scales), size=(K, 1))   # relu-like, mixed magnitudes
    return W, X.astype(np.float32)


def test_three_bf16_pieces_six_products_are_one_fp32_rounding():
    W, X = _operands(0)
    ref = W.astype(np.float64) @ X.astype(np.float64)
    w, a = bf16_pieces(W), bf16_pieces(X)
    assert np.abs(w[0] + w[1] + w[2] - W).max() <= 2.0 ** -22 * np.abs(W).max()          # the split itself: 24 mantissa bits, last bit or two
    got = w[2] @ a[0] + w[0] @ a[2] + w[1] @ a[1] + w[1] @ a[0] + w[0] @ a[1] + w[0] @ a[0]
    noise = np.abs((W @ X).astype(np.float64) - ref)                                      # an fp32 matrix product's own accumulation noise
    err = np.abs(got - ref)
    assert err.max() < noise.max() and err.mean() < 0.5 * noise.mean(), (err.max(), noise.max())


def test_two_fp16_pieces_three_products_are_one_fp32_rounding():
    W, X = _operands(1)
    ref = W.astype(np.float64) @ X.astype(np.float64)
    w, a = fp16_pieces(W, 256.0), fp16_pieces(X)                                          # layout.h H2_WSCALE; activations unscaled
    got = (w[1] @ a[0] + w[0] @ a[1] + w[0] @ a[0]) / 256.0
    noise = np.abs((W @ X).astype(np.float64) - ref)
    err = np.abs(got - ref)
    assert err.max() < noise.max() and err.mean() < 0.75 * noise.mean(), (err.max(), noise.max(), err.mean(), noise.mean())
    # what the pack scale is for: unscaled weights of this size have SUBNORMAL second pieces, and the split is several times worse
    wu = fp16_pieces(W, 1.0)
    err_u = np.abs(wu[1] @ a[0] + wu[0] @ a[1] + wu[0] @ a[0] - ref)
    assert err_u.mean() > 2.0 * err.mean()


def test_fp16_split_precision_by_magnitude():
    """|x| >= 2^-2 after scaling: both pieces normal, relative error <= 2^-22 (two roundings of 2^-12 each, with slack); below: an ABSOLUTE error
    <= 2^-25 (half a subnormal step) -- the bound DESIGN 3.4c quotes; out of range: non-finite pieces, which the pack refuses / the core poisons."""
    rng = np.random.default_rng(2)
    big = (rng.uniform(0.25, 60000.0, 100000) * rng.choice([-1.0, 1.0], 100000)).astype(np.float32)
    h = fp16_pieces(big)
    assert (np.abs(h[0] + h[1] - big) <= 2.0 ** -22 * np.abs(big)).all()
    small = (rng.uniform(0.0, 0.25, 100000) * rng.choice([-1.0, 1.0], 100000)).astype(np.float32)
    h = fp16_pieces(small)
    assert (np.abs(h[0] + h[1] - small) <= 2.0 ** -25).all()
    over = fp16_pieces(np.array([70000.0, -1e6], np.float32))
    assert not np.isfinite(over[0]).any()


def test_sincos_pow2_reduction_is_accurate_in_fp32():
    """csrc/sincos_pow2.h (the embeddings' sin / cos of 2^k x since round 4), restated operation for operation in numpy float32 (fma = one rounding
    of the exact product-sum): against float64 over x in [-6, 6] and the tiny-coordinate range, k = 0 .. 16, the error stays below 1.1e-7 --
    ~1.3 ulp of a value in [0.5, 1) -- with a mean of 1.5e-8; ocml's sincosf, which it replaces, is specified to 2 ulp."""
    f32 = np.float32

    def fma(a, b, c):
        return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)
    c_hi, c_lo, tp_hi, tp_lo = f32(0.159154937), f32(6.42063824e-09), f32(6.28318548), f32(-1.74845553e-07)
    assert abs(float(c_hi) + float(c_lo) - 1 / (2 * np.pi)) < 1e-16 and abs(float(tp_hi) + float(tp_lo) - 2 * np.pi) < 1e-14

    def sincos(x, k):
        p = (x * c_hi).astype(f32)
        e = (fma(x, c_hi, -p) + (x * c_lo).astype(f32)).astype(f32)
        p4 = (p * f32(2.0 ** (k + 2))).astype(f32)
        j = np.rint(p4).astype(f32)
        y = fma((p4 - j).astype(f32), f32(0.25), (e * f32(2.0 ** k)).astype(f32))
        th = fma(y, tp_hi, (y * tp_lo).astype(f32))
        z = (th * th).astype(f32)
        ps = fma(fma(f32(-1.9515295891e-4), z, f32(8.3321608736e-3)), z, f32(-1.6666654611e-1))
        sn = fma((th * z).astype(f32), ps, th)
        pc = fma(fma(fma(f32(2.443315711809948e-5), z, f32(-1.388731625493765e-3)), z, f32(4.166664568298827e-2)), z, f32(-0.5))
        cs = fma(z, pc, f32(1.0))
        q = j.astype(np.int64) & 3
        s = np.where(q == 0, sn, np.where(q == 1, cs, np.where(q == 2, -sn, -cs)))
        c = np.where(q == 0, cs, np.where(q == 1, -sn, np.where(q == 2, -cs, sn)))
        return s.astype(f32), c.astype(f32)

    rng = np.random.default_rng(0)
    for k in range(17):
        x = np.concatenate([rng.uniform(-6, 6, 200000), rng.uniform(-0.01, 0.01, 20000), [0.0, np.pi, -np.pi, np.pi / 2, 1e-8, 5.0, -5.0]]).astype(f32)
        s, c = sincos(x, k)
        arg = x.astype(np.float64) * 2.0 ** k
        es, ec = np.abs(s - np.sin(arg)), np.abs(c - np.cos(arg))
        assert es.max() <= 1.1e-7 and ec.max() <= 1.1e-7 and es.mean() <= 2e-8 and ec.mean() <= 2e-8, (k, es.max(), ec.max(), es.mean(), ec.mean())
