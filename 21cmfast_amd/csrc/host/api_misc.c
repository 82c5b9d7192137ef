/*
 * api_misc.c -- small exported entry points: the filter test hook, bare FFTs,
 * library management.
 */
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"

enum { WS_MISC_A = 40, WS_MISC_B = 41, WS_MISC_C = 42, WS_MISC_D = 43 };

const char *c21cm_version(void) { return "21cmfast_amd 0.1 (gfx950)"; }
const char *c21cm_last_error(void) { return c21hip_get_error(); }
int c21cm_device_synchronize(void) { return c21hip_device_sync(); }

void c21cm_release_device_cache(void) {
    c21hip_fft_release();
    c21hip_ws_release();
}

int c21cm_fft_r2c(float *box, int nx, int ny, int nz, void *stream) {
    if (!c21hip_is_device_ptr(box)) {
        c21hip_set_error("c21cm_fft_r2c needs a device array");
        return C21CM_VALUE_ERROR;
    }
    return c21hip_fft_r2c(box, nx, ny, nz, stream);
}

int c21cm_fft_c2r(float *box, int nx, int ny, int nz, void *stream) {
    if (!c21hip_is_device_ptr(box)) {
        c21hip_set_error("c21cm_fft_c2r needs a device array");
        return C21CM_VALUE_ERROR;
    }
    return c21hip_fft_c2r(box, nx, ny, nz, stream);
}

/* r2c -> /N -> W(kR) -> c2r.  reference: src/py21cmfast/src/filtering.c:397-445 */
static int filter_common(const float *input, float *out_f32, double *out_f64, int nx, int ny,
                         int nz, double box_len, double box_len_z, int filter_type, double R,
                         double R_param, double R_star, void *stream) {
    const size_t ntot = (size_t)nx * ny * nz;
    const size_t npad = (size_t)nx * ny * 2 * (size_t)(nz / 2 + 1);
    int st;
    const float *d_in = input;
    if (!c21hip_is_device_ptr(input)) {
        float *tmp = (float *)c21hip_ws(WS_MISC_A, ntot * sizeof(float));
        if (!tmp) return C21CM_MEMORY_ALLOC_ERROR;
        if ((st = c21hip_h2d(tmp, input, ntot * sizeof(float), stream))) return st;
        d_in = tmp;
    }
    float *unf = (float *)c21hip_ws(WS_MISC_B, npad * sizeof(float));
    float *fil = (float *)c21hip_ws(WS_MISC_C, npad * sizeof(float));
    if (!unf || !fil) return C21CM_MEMORY_ALLOC_ERROR;
    /* test_filter copies without clipping (filtering.c:411-417) */
    if ((st = c21hip_pack_clip(d_in, unf, nx, ny, nz, 1.0, -1e300, 1e300, stream))) return st;
    if ((st = c21hip_fft_r2c(unf, nx, ny, nz, stream))) return st;
    if ((st = c21hip_divide_inplace_f64(unf, npad, (double)ntot, stream))) return st;
    if (c21hip_fft_is_native(nx, ny, nz)) {
        /* split-layout transform with the window fused into its first pass */
        if ((st = c21hip_padded_to_split(unf, fil, nx, ny, nz, stream))) return st;
        if (filter_type == 5) { /* multiple scattering: (R, R_param) = (R_inner, R_outer) */
            if ((st = c21hip_split_filter_shell(fil, fil, 5, NULL, NULL, 0, 1, nx, ny, nz, box_len,
                                                box_len_z, (float)R, (float)R_param,
                                                (float)R_star, 1, stream)))
                return st;
            if ((st = c21hip_split_z_c2r(fil, unf, 2 * (long)(nz / 2 + 1), nx, ny, nz, stream)))
                return st;
        } else if ((st = c21hip_split_filter_c2r(fil, fil, unf, 2 * (long)(nz / 2 + 1), nx, ny,
                                                 nz, box_len, box_len_z, filter_type, (float)R,
                                                 (float)R_param, 1, stream)))
            return st;
        float *swap = unf;
        unf = fil;
        fil = swap;
    } else {
        if ((st = c21hip_copy_filter_star(unf, fil, nx, ny, nz, box_len, box_len_z, filter_type,
                                          (float)R, (float)R_param, (float)R_star, 1, stream)))
            return st;
        if ((st = c21hip_fft_c2r(fil, nx, ny, nz, stream))) return st;
    }
    if (out_f32) {
        float *d_out = out_f32;
        const int host = !c21hip_is_device_ptr(out_f32);
        if (host) d_out = (float *)c21hip_ws(WS_MISC_A, ntot * sizeof(float));
        if ((st = c21hip_unpack_scale(fil, d_out, nx, ny, nz, 1.0f, stream))) return st;
        if (host && (st = c21hip_d2h(out_f32, d_out, ntot * sizeof(float), stream))) return st;
    }
    if (out_f64) {
        float *dense = (float *)c21hip_ws(WS_MISC_A, ntot * sizeof(float));
        if ((st = c21hip_unpack_scale(fil, dense, nx, ny, nz, 1.0f, stream))) return st;
        double *d_out = out_f64;
        const int host = !c21hip_is_device_ptr(out_f64);
        if (host) d_out = (double *)c21hip_ws(WS_MISC_D, ntot * sizeof(double));
        if (!d_out) return C21CM_MEMORY_ALLOC_ERROR;
        if ((st = c21hip_widen(dense, d_out, ntot, stream))) return st;
        if (host && (st = c21hip_d2h(out_f64, d_out, ntot * sizeof(double), stream))) return st;
    }
    return c21hip_sync(stream);
}

int c21cm_filter_grid(const float *input, float *output, int nx, int ny, int nz, double box_len,
                      double box_len_z, int filter_type, double R, double R_param, void *stream) {
    if (!input || !output || nx < 1 || ny < 1 || nz < 2) return C21CM_VALUE_ERROR;
    return filter_common(input, output, NULL, nx, ny, nz, box_len, box_len_z, filter_type, R,
                         R_param, 0., stream);
}

/* The exported test hook of the reference ABI; geometry from the broadcast globals. */
int test_filter(float *input_box, double R, double R_param, double R_star, int filter_flag,
                double *result) {
    if (!simulation_options_global) {
        c21hip_set_error("test_filter: Broadcast_struct_global_all has not been called");
        return C21CM_VALUE_ERROR;
    }
    const SimulationOptions *so = simulation_options_global;
    const int n = so->HII_DIM;
    const int nz = (int)(so->NON_CUBIC_FACTOR * so->HII_DIM);
    const float len_z = so->BOX_LEN * so->NON_CUBIC_FACTOR; /* float product, filtering.c:313 */
    /* R_star: only the multiple-scattering window (type 5) uses it */
    return filter_common(input_box, NULL, result, n, n, nz, (double)so->BOX_LEN, (double)len_z,
                         filter_flag, R, R_param, R_star, NULL);
}
