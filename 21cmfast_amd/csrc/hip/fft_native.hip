// fft_native.hip -- hand-written power-of-two 3-D real FFT for gfx950 (placeholder:
// reports "unsupported" so fft.hip routes every size through rocFFT).
#include <hip/hip_runtime.h>

#include "c21hip.h"
#include "c21cm_abi.h"

extern "C" int c21hip_native_fft_supported(int nx, int ny, int nz) {
    (void)nx;
    (void)ny;
    (void)nz;
    return 0;
}
extern "C" int c21hip_native_fft_r2c(float *padded, int nx, int ny, int nz, void *stream) {
    (void)padded; (void)nx; (void)ny; (void)nz; (void)stream;
    return C21CM_VALUE_ERROR;
}
extern "C" int c21hip_native_fft_c2r(float *padded, int nx, int ny, int nz, void *stream) {
    (void)padded; (void)nx; (void)ny; (void)nz; (void)stream;
    return C21CM_VALUE_ERROR;
}
