"""The arithmetic of csrc/wino_gemm_split.hip restated in numpy (no GPU): an f32 number as three bf16 terms, a product as the six
largest term products.  Checks the bounds the kernel's header states, on random and on adversarial operands."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32: what v_cvt_pk_bf16_f32 does for finite inputs"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return ((u >> 16) << 16).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r1 = x - h            # exact in f32
    m = bf16_rne(r1)
    r2 = r1 - m           # exact in f32
    return h, m, bf16_rne(r2)


def rand_operands(rng, n):
    mant = rng.uniform(1.0, 2.0, n)
    expo = rng.integers(-20, 20, n)
    sign = rng.choice([-1.0, 1.0], n)
    return (sign * mant * 2.0 ** expo).astype(np.float32)


def test_three_terms_rebuild_the_number():
    rng = np.random.default_rng(0)
    x = np.concatenate([rand_operands(rng, 200000), np.float32([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 3.0e38, 1.0e-30,
                                                                   np.pi, -np.e, 255.0 / 256.0, 1.0 + 2.0 ** -8 + 2.0 ** -16])])
    h, m, l = split3(x)
    x64 = x.astype(np.float64)
    # the subtractions are exact: r1, r2 computed in f32 equal the f64 differences
    assert np.array_equal((x - h).astype(np.float64), x64 - h.astype(np.float64))
    assert np.array_equal(((x - h) - m).astype(np.float64), x64 - h.astype(np.float64) - m.astype(np.float64))
    rest = np.abs(x64 - (h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)))
    assert np.all(rest <= 2.0 ** -27 * np.abs(x64))
    # most f32 numbers are rebuilt exactly (24 significant bits fit in 3 x 8 plus the signs)
    assert (rest == 0).mean() > 0.9
    # term sizes: |m| <= 2^-8 |h|-ish, |l| <= 2^-16
    assert np.all(np.abs(m) <= 2.0 ** -8 * np.abs(x) * (1 + 2.0 ** -7)) and np.all(np.abs(l) <= 2.0 ** -17 * np.abs(x) * (1 + 2.0 ** -6))


def test_six_products_are_a_product_to_f32_accuracy():
    rng = np.random.default_rng(1)
    x, y = rand_operands(rng, 300000), rand_operands(rng, 300000)
    xs = [t.astype(np.float64) for t in split3(x)]
    ys = [t.astype(np.float64) for t in split3(y)]
    six = xs[0] * ys[0] + (xs[0] * ys[1] + xs[1] * ys[0]) + (xs[0] * ys[2] + xs[1] * ys[1] + xs[2] * ys[0])
    exact = x.astype(np.float64) * y.astype(np.float64)
    err = np.abs(six - exact) / np.abs(exact)
    assert err.max() <= 3.5 * 2.0 ** -26, err.max()       # the dropped terms m l' + l m' + l l' and the two rests
    assert err.max() < 2.0 ** -24                         # below half an ulp of the f32 product itself
    # every term product is exact in f32 (8 x 8 significant bits): the matrix pipe adds exact numbers
    for a in xs:
        for b in ys:
            p = a * b
            assert np.array_equal(p.astype(np.float32).astype(np.float64), p) or np.all(np.abs(p[p.astype(np.float32) != p]) < 1e-37)


def test_dot_products_match_f32_accumulation():
    """K = 1024 dot products: six-product sums accumulated in f32 against f64 -- the error is that of an f32 accumulation, the same as
    with exact f32 products (what v_mfma_f32_32x32x2_f32 does)"""
    rng = np.random.default_rng(2)
    K, n = 1024, 2000
    x = rng.standard_normal((n, K)).astype(np.float32)
    y = rng.standard_normal((n, K)).astype(np.float32)
    exact = (x.astype(np.float64) * y.astype(np.float64)).sum(1)
    xs, ys = split3(x), split3(y)
    acc6 = np.zeros(n, np.float32)
    accf = np.zeros(n, np.float32)
    order = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]           # the kernel's order within a k-step
    for k0 in range(0, K, 16):
        for ta, tb in order:
            part = (xs[ta][:, k0:k0 + 16].astype(np.float64) * ys[tb][:, k0:k0 + 16].astype(np.float64)).sum(1)
            acc6 = (acc6.astype(np.float64) + part).astype(np.float32)
        partf = (x[:, k0:k0 + 16].astype(np.float64) * y[:, k0:k0 + 16].astype(np.float64)).sum(1)
        accf = (accf.astype(np.float64) + partf).astype(np.float32)
    scale = np.sqrt(K)
    e6 = np.abs(acc6 - exact).max() / scale
    ef = np.abs(accf - exact).max() / scale
    assert e6 < 4e-6 and e6 < 4 * ef + 1e-7, (e6, ef)
