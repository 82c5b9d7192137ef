// split_layout.h -- index maps of the x-blocked split k-space layout (fft_native.hip: the main block of
// a spectrum with nx >= 1024 is stored [x / XB][y][x % XB][k_z], XB = 2^lb; lb = 0: plain [x][y][k_z];
// the Nyquist plane is always [x][y]).  For the element-wise kernels that walk a spectrum in memory
// order and need the wavenumbers of a line, or look a logical line up.
#pragma once
// memory line m -> logical line x * ny + y
__host__ __device__ __forceinline__ long c21_logical_line(long m, int ny, int lb) {
    if (lb == 0) return m;
    const long blk = (long)ny << lb;
    const long xbk = m / blk, rem = m - xbk * blk;
    const long y = rem >> lb, xi = rem & ((1 << lb) - 1);
    return ((xbk << lb) + xi) * ny + y;
}
// logical line x * ny + y -> memory line (x / XB * ny + y) * XB + x % XB
__host__ __device__ __forceinline__ long c21_memory_line(long l, int ny, int lb) {
    if (lb == 0) return l;
    const long x = l / ny, y = l - x * ny;
    return (((x >> lb) * ny + y) << lb) + (x & ((1 << lb) - 1));
}
