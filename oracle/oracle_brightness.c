/*
 * oracle_brightness.c -- CPU restatement of ComputeBrightnessTemp's per-cell sweep.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  reference:
 * src/py21cmfast/src/BrightnessTemperatureBox.c:58-97 -- float products left to right, the
 * spin-temperature branch in double, the box mean as a double sum divided by (float)N.
 */
#include <math.h>
#include <stddef.h>

#include "oracle.h"

int oracle_brightness_grids(const c21cm_brightness_spec *s, const float *density,
                            const float *neutral_fraction, const float *spin_temperature,
                            float *brightness_temp, float *tau_21, double *mean_out) {
    if (!s || !density || !neutral_fraction || !brightness_temp) return C21CM_VALUE_ERROR;
    if (s->use_ts_fluct && (!spin_temperature || !tau_21)) return C21CM_VALUE_ERROR;
    const float const_factor = s->const_factor, T_rad = s->T_rad;
    const double redshift = s->redshift;
    double ave = 0.;
#pragma omp parallel for reduction(+ : ave)
    for (size_t i = 0; i < s->n_cells; i++) {
        const float pixel_deltax = density[i], pixel_x_HI = neutral_fraction[i];
        float bt = const_factor * pixel_x_HI * (1 + pixel_deltax);
        if (s->use_ts_fluct) {
            bt *= (1. + redshift) / (1000. * spin_temperature[i]);
            tau_21[i] = bt;
            bt = (1. - exp(-bt)) * 1000. * (spin_temperature[i] - T_rad) / (1. + redshift);
        }
        brightness_temp[i] = bt;
        ave += bt;
    }
    if (!isfinite(ave)) return C21CM_INFINITY_OR_NAN_ERROR;
    if (mean_out) *mean_out = ave / (float)s->n_cells;
    return 0;
}
