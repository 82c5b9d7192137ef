/*
 * brightness_driver.c -- C host driver of the ComputeBrightnessTemp grid algorithm
 * (src/py21cmfast/src/BrightnessTemperatureBox.c:22-105): one per-cell sweep and a sum.
 * Host arrays are staged through workspace slots, device arrays are used in place.
 */
#include <math.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"

enum { WS_BT_DENS = 36, WS_BT_XH, WS_BT_TS, WS_BT_OUT, WS_BT_TAU = 44, WS_BT_PART = 45 };

#define TRY(expr)         \
    do {                  \
        int st_ = (expr); \
        if (st_) {        \
            status = st_; \
            goto done;    \
        }                 \
    } while (0)

static const float *bt_in(int slot, const float *p, size_t bytes, void *stream, int *status) {
    if (!p || *status || c21hip_is_device_ptr(p)) return p;
    void *d = c21hip_ws(slot, bytes);
    if (!d) {
        *status = C21CM_MEMORY_ALLOC_ERROR;
        return NULL;
    }
    *status = c21hip_h2d(d, p, bytes, stream);
    return (const float *)d;
}

int c21cm_brightness_grids(const c21cm_brightness_spec *s, const float *density,
                           const float *neutral_fraction, const float *spin_temperature,
                           float *brightness_temp, float *tau_21, double *mean_out, void *stream) {
    int status = 0;
    if (!s || !density || !neutral_fraction || !brightness_temp || s->n_cells == 0) {
        c21hip_set_error("brightness: density, neutral_fraction and brightness_temp are required");
        return C21CM_VALUE_ERROR;
    }
    if (s->use_ts_fluct && (!spin_temperature || !tau_21)) {
        c21hip_set_error("brightness: USE_TS_FLUCT needs spin_temperature and tau_21");
        return C21CM_VALUE_ERROR;
    }
    const size_t bytes = s->n_cells * sizeof(float);
    const float *d_dens = bt_in(WS_BT_DENS, density, bytes, stream, &status);
    const float *d_xh = bt_in(WS_BT_XH, neutral_fraction, bytes, stream, &status);
    const float *d_ts = s->use_ts_fluct ? bt_in(WS_BT_TS, spin_temperature, bytes, stream, &status) : NULL;
    if (status) return status;
    const int host_bt = !c21hip_is_device_ptr(brightness_temp);
    const int host_tau = s->use_ts_fluct && !c21hip_is_device_ptr(tau_21);
    float *d_bt = host_bt ? (float *)c21hip_ws(WS_BT_OUT, bytes) : brightness_temp;
    float *d_tau = !s->use_ts_fluct ? NULL : (host_tau ? (float *)c21hip_ws(WS_BT_TAU, bytes) : tau_21);
    double *part = (double *)c21hip_ws(WS_BT_PART, (C21HIP_PARTIALS + 8) * sizeof(double));
    if (!d_bt || (s->use_ts_fluct && !d_tau) || !part) return C21CM_MEMORY_ALLOC_ERROR;
    double *sum_dev = part + C21HIP_PARTIALS;
    TRY(c21hip_brightness_temp(d_dens, d_xh, d_ts, d_bt, d_tau, s->n_cells, s->const_factor, s->T_rad,
                               s->redshift, s->use_ts_fluct, part, sum_dev, stream));
    double sum = 0.;
    TRY(c21hip_d2h(&sum, sum_dev, sizeof(double), stream));
    if (host_bt) TRY(c21hip_d2h(brightness_temp, d_bt, bytes, stream));
    if (host_tau) TRY(c21hip_d2h(tau_21, d_tau, bytes, stream));
    TRY(c21hip_sync(stream));
    if (!isfinite(sum)) { /* BrightnessTemperatureBox.c:92-95 */
        c21hip_set_error("brightness: average brightness temperature is infinite or NaN");
        status = C21CM_INFINITY_OR_NAN_ERROR;
        goto done;
    }
    if (mean_out) *mean_out = sum / (float)s->n_cells; /* :97 */
done:
    return status;
}
