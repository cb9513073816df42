"""GPU parity: HIP NTT / iNTT / coset extension (through the C ABI) == oracle restatement of halo2's
best_fft / EvaluationDomain, bit for bit, plus size-independent properties at BASELINE sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
