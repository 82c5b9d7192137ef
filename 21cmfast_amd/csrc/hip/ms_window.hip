// ms_window.hip -- host-side exports of the multiple-scattering window's helper functions, the
// three the reference exposes to its Python tests.
// reference: src/py21cmfast/src/filtering.c:126-160,258-293; _functionprototypes_wrapper.h:134-136
#include "ms_window.h"

extern "C" double compute_mu_for_multiple_scattering(double x_em) { return ms_mu(x_em); }
extern "C" double compute_eta_for_multiple_scattering(double x_em) { return ms_eta(x_em); }

extern "C" double hyper_2F3(double kR, double alpha, double beta) {
    MsSide s{};
    s.alpha = alpha;
    s.beta = beta;
    if (beta != 0.) ms_side_fill(s, alpha, beta);  // beta == 0: the straight-line window
    return ms_hyper(kR, s);
}
