"""Host random streams of the library (nuzero_amd/csrc/rng_host.cpp) against
numpy's legacy RandomState: golden vectors and live draws.  CPU only."""
import ctypes

import numpy as np
import pytest

from nuzero_amd._lib import lib


class Rng:
    def __init__(self, seed):
        self.h = lib.nz_rng_create(seed)

    def __del__(self):
        lib.nz_rng_destroy(self.h)

    def random_sample(self):
        return lib.nz_rng_double(self.h)

    def gamma(self, a, b, n):
        out = (ctypes.c_double * max(n, 1))()
        lib.nz_rng_gamma(self.h, a, b, n, out)
        return list(out)[:n]


def test_raw_and_double_golden(rng_kat):
    r = Rng(42)
    raw = np.array([lib.nz_rng_u32(r.h) for _ in range(1300)], np.uint32)
    assert np.array_equal(raw, rng_kat["raw_u32_seed42"])
    r = Rng(42)
    dbl = np.array([r.random_sample() for _ in range(1300)])
    assert np.array_equal(dbl, rng_kat["double_seed42"])


def test_draw_pattern_golden(rng_kat):
    """gamma(a,b,n); random(); random(); choice(9,p) -- the per-move pattern."""
    p = np.arange(1, 10, dtype=np.float64)
    p /= p.sum()
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    for key, want in rng_kat.items():
        if not key.startswith("s"):
            continue
        seed, a, b = key.split("_")
        r = Rng(int(seed[1:]))
        a, b = float(a[1:]), float(b[1:])
        got = []
        for n in (9, 0, 8, 1, 7, 40):
            got.extend(r.gamma(a, b, n))
            got.append(r.random_sample())
            got.append(r.random_sample())
            got.append(float(np.searchsorted(cdf, r.random_sample(), side="right")))
        assert np.array_equal(np.array(got), want), key


@pytest.mark.parametrize("alpha,beta", [(0.15, 1.0), (0.03, 1.0), (0.5, 3.0), (1.0, 0.7), (2.0, 1.0), (7.5, 0.1)])
def test_gamma_live(alpha, beta):
    for seed in (3, 99, 2 ** 32 - 1):
        r = Rng(seed)
        rs = np.random.RandomState(seed)
        for n in (1, 5, 9, 33):
            assert r.gamma(alpha, beta, n) == rs.gamma(alpha, beta, n).tolist()
            assert r.random_sample() == rs.random_sample()


def test_reseed():
    r = Rng(5)
    first = [r.random_sample() for _ in range(10)]
    lib.nz_rng_seed(r.h, 5)
    assert [r.random_sample() for _ in range(10)] == first


def test_pairwise_order():
    """The summation order the device code assumes for np.sum (n = 9 and the
    general rule) really is numpy's."""
    def pw(a):
        n = len(a)
        if n < 8:
            r = a.dtype.type(0)
            for x in a:
                r = r + x
            return r
        if n <= 128:
            r = [a[j] for j in range(8)]
            i = 8
            while i < n - n % 8:
                for j in range(8):
                    r[j] = r[j] + a[i + j]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < n:
                res = res + a[i]
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return pw(a[:n2]) + pw(a[n2:])
    rs = np.random.RandomState(0)
    for dt in (np.float64, np.float32):
        for n in (3, 7, 8, 9, 17, 130, 525):
            for _ in range(50):
                a = (rs.uniform(0, 1, n) ** 8).astype(dt)
                assert np.sum(a) == pw(a)
