"""The arithmetic claim behind the library's bf16x6 mode (include/rlt_hip.h RLT_PRECISION_BF16X6; csrc/gemm.hip split4x3,
csrc/attention6.hip split4x3_6), restated in numpy and checked on the CPU:

  * every fp32 value splits EXACTLY into three bf16 values, x = h + m + l with h = bf16(x), m = bf16(x - h),
    l = x - h - m (round-to-nearest-even conversions, the residuals formed in fp32) - over the whole fp32 range except
    its two ends: above 3.39e38 bf16(x) rounds to infinity, and below ~1e-33 the last residual falls into bf16's
    denormal spacing (an absolute error below 1e-40 - nothing on this path lives there);
  * each of the nine partial products of two such splits is exact in fp32 (8 x 8 significand bits), and the three the
    kernels drop (m*l' + l*m' + l*l') are together below 2^-23 |a*b| in the worst case (|m| <= 2^-8 |x|, |l| <= 2^-16 |x|) -
    under one fp32 ulp of the product; over 3 * 10^5 random pairs the largest is 2^-24.3, the median 2^-29.
"""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    lo = bf16_rne(r2)
    return h, m, lo, r2


def _samples():
    rs = np.random.RandomState(20240)
    vals = [rs.standard_normal(200000).astype(np.float32),
            (rs.standard_normal(200000) * 1e-6).astype(np.float32),
            (rs.standard_normal(200000) * 1e6).astype(np.float32),
            np.exp(rs.uniform(-60, 60, 200000)).astype(np.float32) * rs.choice([-1, 1], 200000).astype(np.float32),
            rs.randint(0, 2 ** 24, 200000).astype(np.float32),                      # all 24 significand bits in use
            np.array([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.0 + 2.0 ** -16, 3.3e38, 1e-30], dtype=np.float32)]
    return np.concatenate(vals)


def test_three_way_bf16_split_is_exact():
    x = _samples()
    h, m, lo, r2 = split3(x)
    assert np.array_equal(lo, r2)                                   # the last residual is representable: nothing is rounded away
    total = h.astype(np.float64) + m.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64))              # h + m + l == x exactly
    ax = np.abs(x.astype(np.float64))
    nz = ax > 0
    assert (np.abs(m.astype(np.float64))[nz] <= ax[nz] * 2.0 ** -8).all()      # |m| <= 2^-8 |x|  (half a bf16 ulp)
    assert (np.abs(lo.astype(np.float64))[nz] <= ax[nz] * 2.0 ** -16).all()    # |l| <= 2^-16 |x|


def test_six_products_carry_the_product_to_under_one_ulp():
    rs = np.random.RandomState(7)
    x = _samples()
    a = x[rs.permutation(x.size)[:300000]]
    b = x[rs.permutation(x.size)[:300000]]
    keep = (np.abs(a.astype(np.float64) * b.astype(np.float64)) < 1e37) & (np.abs(a.astype(np.float64) * b.astype(np.float64)) > 1e-30)
    a, b = a[keep], b[keep]
    ah, am, al, _ = split3(a)
    bh, bm, bl, _ = split3(b)
    f8 = lambda v: v.astype(np.float64)
    # every partial product is exact in fp32 (what one bf16 MFMA multiplies): check in float32 against float64
    for p, q in ((ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm)):
        assert np.array_equal((p * q).astype(np.float64), f8(p) * f8(q))
    six = f8(ah) * f8(bh) + f8(ah) * f8(bm) + f8(am) * f8(bh) + f8(ah) * f8(bl) + f8(al) * f8(bh) + f8(am) * f8(bm)
    exact = f8(a) * f8(b)
    dropped = np.abs(exact - six)
    rel = dropped / np.abs(exact)
    assert (rel <= 2.0 ** -23).all()                                 # the bound
    assert rel.max() <= 2.0 ** -24 and np.median(rel) <= 2.0 ** -28   # what random operands show (2^-24.3, 2^-29)
    # for comparison: the two-way split of the bf16x3 mode keeps 16 bits (hi*hi + hi*lo + lo*hi with lo = bf16(x - hi))
    al2, bl2 = bf16_rne(a - ah), bf16_rne(b - bh)
    three = f8(ah) * f8(bh) + f8(ah) * f8(bl2) + f8(al2) * f8(bh)
    assert np.percentile(np.abs(exact - three) / np.abs(exact), 99) > 2.0 ** -20
