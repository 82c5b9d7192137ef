"""Pins of the GSL random-stream restatement (oracle/oracle_gslrng.c) and of the product's own
host implementation (csrc/host/gsl_stream.c) -- CPU only.

* mt19937 word for word against numpy's MT19937 bit generator with legacy seeding (the same
  2002 init_genrand GSL uses; reference call site: src/py21cmfast/src/rng.c:33-35).
* gfsr4: the defining four-tap recurrence holds on the emitted words, and the generator is
  equidistributed enough to pass moment checks.
* per-thread seeds of seed_rng_threads(12345) for N_THREADS = 1, 2 (rng.c:36-56) are frozen: the
  reference's fixtures (tests/test_reference_fixtures.py) only come out right with exactly these.
* the product's host stream equals the oracle's bit for bit (two independent implementations).
"""

import ctypes as C
import importlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def olib(oracle):
    lib = oracle.load()
    lib.oracle_gsl_rng_alloc.restype = C.c_void_p
    lib.oracle_gsl_rng_alloc.argtypes = [C.c_int, C.c_ulong]
    lib.oracle_gsl_rng_get.restype = C.c_uint
    lib.oracle_gsl_rng_get.argtypes = [C.c_void_p]
    lib.oracle_gsl_rng_free.argtypes = [C.c_void_p]
    lib.oracle_gsl_ran_ugaussian.restype = C.c_double
    lib.oracle_gsl_ran_ugaussian.argtypes = [C.c_void_p]
    lib.oracle_gsl_thread_seeds.argtypes = [C.c_ulonglong, C.c_int, C.c_void_p]
    lib.oracle_gsl_mode_deviates.argtypes = [C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p]
    return lib


@pytest.mark.parametrize("seed", [12345, 1, 0, 4357, 2**32 + 7])
def test_mt19937_matches_numpy_legacy_seeding(olib, seed):
    r = olib.oracle_gsl_rng_alloc(0, seed)
    mine = np.array([olib.oracle_gsl_rng_get(r) for _ in range(3000)], dtype=np.uint64)
    olib.oracle_gsl_rng_free(r)
    bg = np.random.MT19937()
    # gsl_rng_set: seed 0 means 4357, and only the low 32 bits enter the state
    bg._legacy_seeding((seed & 0xFFFFFFFF) or 4357)
    np.testing.assert_array_equal(mine, bg.random_raw(3000).astype(np.uint64))


def test_gfsr4_recurrence_and_moments(olib):
    r = olib.oracle_gsl_rng_alloc(1, 105396548)
    w = np.array([olib.oracle_gsl_rng_get(r) for _ in range(40000)], dtype=np.uint32)
    olib.oracle_gsl_rng_free(r)
    n = np.arange(9689, len(w))
    np.testing.assert_array_equal(w[n], w[n - 471] ^ w[n - 1586] ^ w[n - 6988] ^ w[n - 9689])
    u = w / 2.0**32
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 2e-3
    assert len(np.unique(w)) > 39990


def test_thread_seeds_are_frozen(olib):
    s = (C.c_uint * 2)()
    assert olib.oracle_gsl_thread_seeds(12345, 1, s) == 0
    assert s[0] == 26861751
    assert olib.oracle_gsl_thread_seeds(12345, 2, s) == 0
    assert list(s) == [26861751, 105396548]


def test_polar_gaussian_moments(olib):
    r = olib.oracle_gsl_rng_alloc(0, 99)
    x = np.array([olib.oracle_gsl_ran_ugaussian(r) for _ in range(200000)])
    olib.oracle_gsl_rng_free(r)
    assert abs(x.mean()) < 8e-3 and abs(x.std() - 1) < 6e-3 and abs((x**4).mean() - 3) < 0.06


def test_generators_reproduce_gsl_self_test_values(olib, pkg):
    """GSL's rng/test.c: the n-th output after gsl_rng_set(seed) of each generator upstream's
    threads use.  Both restatements (oracle: 64-bit modular arithmetic; library: Schrage's
    decomposition as GSL codes it) must give them."""
    lib = pkg.load()
    lib.c21_gsl_nth_output.restype = C.c_ulong
    lib.c21_gsl_nth_output.argtypes = [C.c_int, C.c_ulong, C.c_int]
    olib.oracle_gsl_rng_alloc.restype = C.c_void_p
    olib.oracle_gsl_rng_alloc.argtypes = [C.c_int, C.c_ulong]
    olib.oracle_gsl_rng_get.restype = C.c_uint32
    olib.oracle_gsl_rng_get.argtypes = [C.c_void_p]
    olib.oracle_gsl_rng_free.argtypes = [C.c_void_p]
    for kind, seed, n, want in ((0, 4357, 1000, 1186927261), (2, 1, 10000, 719452880),
                                (3, 1, 10000, 2064828650), (4, 1, 10000, 2733957125)):
        assert lib.c21_gsl_nth_output(kind, seed, n) == want, kind
        r = olib.oracle_gsl_rng_alloc(kind, seed)
        v = [olib.oracle_gsl_rng_get(r) for _ in range(n)][-1]
        olib.oracle_gsl_rng_free(r)
        assert v == want, kind
    # seed 0 means the generator's default seed (4357 for mt19937 / gfsr4, 1 for the others)
    assert lib.c21_gsl_nth_output(2, 0, 100) == lib.c21_gsl_nth_output(2, 1, 100)
    assert lib.c21_gsl_nth_output(0, 0, 100) == lib.c21_gsl_nth_output(0, 4357, 100)


@pytest.mark.parametrize("n_threads,shape", [(1, (12, 12, 12)), (2, (15, 15, 15)), (2, (10, 10, 10)),
                                             (3, (10, 10, 10)), (5, (12, 8, 8)), (7, (16, 6, 6))])
def test_product_host_stream_equals_oracle(olib, pkg, n_threads, shape):
    """csrc/host/gsl_stream.c (block-refill word sources) vs oracle/oracle_gslrng.c: the same
    deviates for every mode, including the odd row split of 15 rows over 2 threads."""
    lib = pkg.load()
    lib.c21_gsl_mode_deviates.restype = C.c_int
    lib.c21_gsl_mode_deviates.argtypes = [C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]
    nx, ny, nz = shape
    nzc = nz // 2 + 1
    a = np.zeros((nx, ny, nzc, 2))
    b = np.zeros_like(a)
    assert lib.c21_gsl_mode_deviates(777, n_threads, nx, ny, nzc, a.ctypes.data) == 0
    assert olib.oracle_gsl_mode_deviates(777, n_threads, nx, ny, nz, b.ctypes.data) == 0
    np.testing.assert_array_equal(a, b)
    assert 0.9 < a.std() < 1.1
    # every thread's block is a standard-normal sample of its own generator
    for t in range(n_threads):
        q, rem = divmod(nx, n_threads)
        lo = t * q + min(t, rem)
        blk = a[lo: lo + q + (1 if t < rem else 0)]
        assert abs(blk.mean()) < 6 / np.sqrt(blk.size) and 0.8 < blk.std() < 1.2
    # more threads than rows of modes: the surplus threads of upstream's loop get no iterations
    a2, b2 = np.zeros_like(a), np.zeros_like(a)
    assert lib.c21_gsl_mode_deviates(777, nx + 3, nx, ny, nzc, a2.ctypes.data) == 0
    assert olib.oracle_gsl_mode_deviates(777, nx + 3, nx, ny, nz, b2.ctypes.data) == 0
    np.testing.assert_array_equal(a2, b2)
    assert lib.c21_gsl_mode_deviates(777, 0, nx, ny, nzc, a.ctypes.data) == 3
