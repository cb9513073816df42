"""GPU parity: HIP NTT / iNTT / coset extension (through the C ABI) == oracle restatement of halo2's
best_fft / EvaluationDomain, bit for bit, plus size-independent properties at BASELINE sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=["nine_limb", "eight_limb"])
def limb_form(request, zg):
    """every case on both butterfly back ends (ZG_NTT9: nine 29-bit limbs, the default, and 8 x 32-bit)"""
    before = zg.tuning_get("ZG_NTT9")
    zg.tuning_set("ZG_NTT9", request.param)
    yield request.param
    zg.tuning_set("ZG_NTT9", before)


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 10, 11, 12, 14, 15, 17])
def test_ntt_matches_oracle(ctx, zg, orc, log_n):
    n = 1 << log_n
    a = orc.fill_fr(1000 + log_n, n)
    om, omi = zg.domain_omega(log_n)
    got = ctx.ntt(a, om)
    assert np.array_equal(got, orc.fft(a, om))
    # inverse with divisor == EvaluationDomain::ifft
    div = orc.fr_inv(orc.fr_from_int(n))
    back = ctx.ntt(got, omi, div)
    assert np.array_equal(back, a)


@pytest.mark.parametrize("log_n", [18, 20])
def test_ntt_extended_domain_sizes_match_oracle(ctx, zg, orc, log_n):
    """Full extended-domain sizes (k=15 -> 2^18, k=17 -> 2^20): the WHOLE array against the oracle's best_fft (still
    seconds on the host), then the inverse transform back to the input."""
    n = 1 << log_n
    a = orc.fill_fr(77, n)
    om, omi = zg.domain_omega(log_n)
    got = ctx.ntt(a, om)
    assert np.array_equal(got, orc.fft(a, om))
    div = orc.fr_inv(orc.fr_from_int(n))
    assert np.array_equal(ctx.ntt(got, omi, div), a)


@pytest.mark.parametrize("log_n", [3, 10, 11, 13, 16, 19, 22])
def test_ntt_of_extreme_values(ctx, zg, orc, log_n):
    """The nine-limb butterflies keep sums unreduced between their reductions (csrc/ntt.hip: an only-summed element grows
    fourfold per radix-2^2 group): inputs that drive those sums and differences to their largest magnitudes -- every stored
    value p - 1, p - 1 against 0 in every pairing the stages make, and p - 1 at the even rows of a zero-padded transform."""
    n = 1 << log_n
    top = np.array([(zg.FR_MODULUS - 1 >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    om, omi = zg.domain_omega(log_n)
    div = orc.fr_inv(orc.fr_from_int(n))
    idx = np.arange(n)
    patterns = [np.ones(n, bool)] + [((idx >> b) & 1) == 0 for b in sorted({0, 1, log_n // 2, max(log_n - 2, 0), log_n - 1})]
    for keep in patterns:
        a = np.zeros((n, 4), np.uint64)
        a[keep] = top
        got = ctx.ntt(a, om)
        assert np.array_equal(got, orc.fft(a, om))
        assert np.array_equal(ctx.ntt(got, omi, div), a)
    if 4 <= log_n <= 17:  # coeff_to_extended's zero padding with the largest coefficients
        d = orc.domain(6, log_n - 3) if log_n - 3 >= 1 else None
        if d is not None and d.extended_k == log_n:
            a = np.tile(top, (1 << (log_n - 3), 1))
            assert np.array_equal(ctx.coeff_to_extended(a, log_n - 3, log_n), orc.coeff_to_extended(d, a))


def test_ntt_linearity_and_delta(ctx, zg, orc):
    log_n = 16
    n = 1 << log_n
    om, _ = zg.domain_omega(log_n)
    # NTT(e_1) = (omega^k)_k ; NTT(e_0) = all ones
    e = np.zeros((n, 4), np.uint64)
    e[0] = orc.fr_from_int(1)
    assert (ctx.ntt(e, om) == orc.fr_from_int(1)).all()
    e = np.zeros((n, 4), np.uint64)
    e[1] = orc.fr_from_int(1)
    f = ctx.ntt(e, om)
    w = zg.fr_to_int(om)
    for k in (0, 1, 2, 12345, n - 1):
        assert zg.fr_to_int(f[k]) == pow(w, k, zg.FR_MODULUS)


def test_ntt_batch(ctx, zg, orc):
    log_n, batch = 14, 7
    n = 1 << log_n
    a = np.stack([orc.fill_fr(10 + b, n) for b in range(batch)])
    om, omi = zg.domain_omega(log_n)
    got = ctx.ntt_batch(a, om)
    for b in range(batch):
        assert np.array_equal(got[b], orc.fft(a[b], om))
    div = orc.fr_inv(orc.fr_from_int(n))
    assert np.array_equal(ctx.ntt_batch(got, omi, div), a)


@pytest.mark.parametrize("k,j", [(4, 6), (8, 6), (10, 4), (14, 6), (15, 6), (17, 6)])
def test_coeff_to_extended_and_back(ctx, zg, orc, k, j):
    d = orc.domain(j, k)
    n = 1 << k
    a = orc.fill_fr(50 + k, n)
    ext = ctx.coeff_to_extended(a, k, d.extended_k)
    assert np.array_equal(ext, orc.coeff_to_extended(d, a))
    out_len = n * d.quotient_poly_degree
    h = orc.fill_fr(60 + k, 1 << d.extended_k)
    assert np.array_equal(ctx.extended_to_coeff(h, k, d.extended_k, out_len), orc.extended_to_coeff(d, h))
    back = ctx.extended_to_coeff(ext, k, d.extended_k, out_len)
    assert np.array_equal(back[:n], a) and not back[n:].any()


def test_ntt_rejects_bad_arguments(ctx, zg):
    a = np.zeros((1 << 23, 4), np.uint64)
    om, _ = zg.domain_omega(23)
    with pytest.raises(zg.ZgError) as e:
        ctx.ntt(a, om)
    assert e.value.status == -4  # ZG_ERR_UNSUPPORTED
