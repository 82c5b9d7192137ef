"""Restatement of ``powerbox.get_power(field, boxlength=L, bins_upto_boxlen=True)[:2]``.

TEST INFRASTRUCTURE ONLY.  ``powerbox`` is a third-party dependency of the reference's test
suite (absent from /root/reference and from this image; the reference pins ``powerbox>=0.8``
in pyproject's test extras).  The reference's fixtures under ``tests/test_data/*.h5`` are the
outputs of exactly this call (``tests/produce_integration_test_data.py:367-370,438-446``), so
using them as a pin needs its binning.  Restated from powerbox's published algorithm
(``powerbox.tools.get_power`` / ``angular_average`` / ``_getbins`` / ``_get_binweights``):

* Fourier convention a = b = 1 of ``powerbox.dft.fft``: F(k) = (V / N_tot) * DFT(f), k = 2 pi n / L
  on the fftshift-ed integer grid n = -N/2 .. N/2 - 1;
* P(k) = |F|^2 / V  (``vol_normalised_power=True``);
* number of bins ``int(N / 2.2)``; linear edges from min|k| (= 0: the zero mode is kept,
  ``ignore_zero_mode=False``) to the LARGEST |k| REACHED ALONG AN AXIS (that is what
  ``bins_upto_boxlen=True`` selects instead of the corner of the cube);
* ``np.digitize`` with half-open bins [e_i, e_{i+1}): modes at or beyond the last edge are dropped;
* the power of a bin is the plain mean over all modes of the FULL (not half) Fourier grid in the
  bin, the returned k the mean |k| of those modes (``bin_ave=True``).

Pin: the k vectors stored in every fixture depend on this binning alone (no field data), and
``tests/test_reference_fixtures.py::test_powerbox_binning_reproduces_fixture_k`` holds this
function to them to 1e-12.
"""

from __future__ import annotations

import numpy as np


def get_power(field: np.ndarray, boxlength) -> tuple[np.ndarray, np.ndarray]:
    field = np.asarray(field)
    dim = field.ndim
    N = field.shape
    L = [float(boxlength)] * dim if np.isscalar(boxlength) else [float(x) for x in boxlength]
    V = float(np.prod(L))
    ft = np.fft.fftshift(np.fft.fftn(field.astype(np.float64))) * (V / float(np.prod(N)))
    P = (ft.real**2 + ft.imag**2) / V
    freq = [np.fft.fftshift(np.fft.fftfreq(n, d=l / n)) * 2.0 * np.pi for n, l in zip(N, L)]
    grids = np.meshgrid(*freq, indexing="ij")
    kmag = np.sqrt(sum(g * g for g in grids))
    nbins = int(np.prod(N) ** (1.0 / dim) / 2.2)
    # _getbins with bins_upto_boxlen=True: the smallest of the per-axis maxima of |k|
    max_radius = min(float(np.min(np.max(kmag, axis=i))) for i in range(dim))
    edges = np.linspace(kmag.min(), max_radius, nbins + 1)
    indx = np.digitize(kmag.ravel(), edges)
    counts = np.bincount(indx, minlength=len(edges) + 1)[1:-1]
    with np.errstate(invalid="ignore", divide="ignore"):
        k_av = np.bincount(indx, weights=kmag.ravel(), minlength=len(edges) + 1)[1:-1] / counts
        p_av = np.bincount(indx, weights=P.ravel(), minlength=len(edges) + 1)[1:-1] / counts
    return p_av, k_av


def pdf_histogram(data: np.ndarray, xmin: float, xmax: float, nbins: int):
    """The stair-step PDF of ``produce_perturb_field_data`` (:397-412): density-normalised
    histogram on ``np.linspace(xmin, xmax, nbins)`` edges, each value repeated for the left and
    right edge of its bin."""
    bins, edges = np.histogram(data, bins=np.linspace(xmin, xmax, nbins), range=[xmin, xmax],
                               density=True)
    left, right = edges[:-1], edges[1:]
    X = np.array([left, right]).T.flatten()
    Y = np.array([bins, bins]).T.flatten()
    return X, Y
