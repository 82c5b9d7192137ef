"""The claim the banded barrier of the Eulerian loops rests on (csrc/hip/ionize_kernels.hip:
eul_barrier / eul_threshold): the barrier test of eulerian_mask_kernel -- IonisationBox.c:1066-1118 with
the mean fix of :1022-1027,

    c = mf * (double) f;  if (mass_dep_zeta && c < f_limit) c = f_limit;  crossed = c * zeta > 1 - x_e

-- is monotone in the float f >= 0 and in the mean fix mf >= 0 under IEEE double rounding, so that
  (i)  for a given mf (and x_e = 0) it IS a threshold on f: the smallest float that passes, found by
       bisection on the bit pattern, decides every other float by one comparison, and
  (ii) a cell on which both ends of a band [mf_lo, mf_hi] agree is decided for every mf inside the band.
Restated in numpy (the same IEEE operations, no contraction: the library is built with
-ffp-contract=off) and checked on random and adversarial inputs.  CPU only."""

import numpy as np
import pytest


def barrier(mf, f, zeta, mass_dep_zeta=False, f_limit=0.0, xe=0.0):
    c = np.float64(mf) * np.asarray(f, np.float32).astype(np.float64)
    if mass_dep_zeta:
        c = np.where(c < f_limit, np.float64(f_limit), c)
    return c * np.float64(zeta) > (1.0 - np.float64(xe))


def threshold(mf, zeta, mass_dep_zeta=False, f_limit=0.0):
    """eul_threshold: smallest non-negative float that passes (+inf: none, 0: all)."""
    as_f = lambda b: np.array([b], np.uint32).view(np.float32)[0]
    if barrier(mf, np.float32(0.0), zeta, mass_dep_zeta, f_limit):
        return np.float32(0.0)
    lo, hi = 0, 0x7F800000
    with np.errstate(over="ignore", invalid="ignore"):
        if not barrier(mf, as_f(hi), zeta, mass_dep_zeta, f_limit):
            return as_f(hi)
        while hi - lo > 1:
            mid = lo + (hi - lo) // 2
            if barrier(mf, as_f(mid), zeta, mass_dep_zeta, f_limit):
                hi = mid
            else:
                lo = mid
    return as_f(hi)


@pytest.mark.parametrize("mass_dep_zeta,f_limit", [(False, 0.0), (True, 1e-4), (True, 0.2)])
def test_barrier_is_a_threshold_on_the_float_f_coll(mass_dep_zeta, f_limit):
    rng = np.random.default_rng(3)
    for _ in range(40):
        mf = float(np.exp(rng.uniform(np.log(0.2), np.log(50.0))))
        zeta = float(np.exp(rng.uniform(np.log(3.0), np.log(200.0))))
        T = threshold(mf, zeta, mass_dep_zeta, f_limit)
        # random f_coll values over many decades, plus the floats around the threshold
        f = np.exp(rng.uniform(np.log(1e-12), np.log(2.0), 20000)).astype(np.float32)
        if np.isfinite(T) and T > 0:
            bits = np.array([T], np.float32).view(np.uint32)[0]
            near = (np.arange(-200, 201) + int(bits)).astype(np.uint32).view(np.float32)
            f = np.concatenate([f, near])
        assert np.array_equal(barrier(mf, f, zeta, mass_dep_zeta, f_limit), f >= T), (mf, zeta, T)
        if mass_dep_zeta and f_limit * zeta > 1:
            assert T == 0  # the floor alone ionises: every cell crosses
        elif np.isfinite(T):
            assert T > 0 and abs(float(T) * mf * zeta - 1.0) < 1e-6


def test_cells_both_ends_of_a_band_agree_on_are_decided_for_every_mean_fix_inside():
    rng = np.random.default_rng(4)
    zeta = 31.7
    for mass_dep_zeta, f_limit in ((False, 0.0), (True, 3e-3)):
        for _ in range(20):
            mf_lo = float(np.exp(rng.uniform(np.log(0.5), np.log(20.0))))
            mf_hi = mf_lo * (1.0 + rng.uniform(1e-6, 0.08))
            f = np.exp(rng.uniform(np.log(1e-6), np.log(1.0), 50000)).astype(np.float32)
            xe = np.clip(rng.uniform(-0.05, 0.5, f.size), 0.0, 0.999).astype(np.float32)  # the x_e variant
            lo = barrier(mf_lo, f, zeta, mass_dep_zeta, f_limit, xe)
            hi = barrier(mf_hi, f, zeta, mass_dep_zeta, f_limit, xe)
            assert not np.any(lo & ~hi)  # monotone in mf: sure => maybe
            for mf in np.concatenate([[mf_lo, mf_hi], rng.uniform(mf_lo, mf_hi, 6)]):
                ex = barrier(mf, f, zeta, mass_dep_zeta, f_limit, xe)
                assert np.array_equal(ex[lo], np.ones(lo.sum(), bool))      # decided "crosses": crosses
                assert not np.any(ex[~hi])                                  # decided "does not": does not
            assert 0 < np.count_nonzero(hi & ~lo) < 0.1 * f.size            # the undecided sliver exists, and is one
