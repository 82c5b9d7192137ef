// fft_device.h -- device-side building blocks shared by the transform kernels of fft_native.hip and
// the plane-fused pass Y + Z of plane_yz.hip: complex helpers, the small DFTs in registers and the
// wave-level complex-to-real line transform.  Included INSIDE each translation unit's anonymous
// namespace; the arithmetic of a line is the same instruction sequence wherever it is instantiated
// (-ffp-contract=off), which is what keeps the fused and the unfused passes bit-identical.
#pragma once
#ifndef C21X_ZW_DPP
#define C21X_ZW_DPP 0
#endif

// ------------------------------------------------------------------ complex helpers
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// multiply by +i (SIGN > 0) or -i (SIGN < 0)
template <int SIGN>
__device__ __forceinline__ float2 mul_i(float2 a) {
    return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// Small DFTs, y_j = sum_k a_k exp(SIGN * 2 pi i j k / R), outputs in natural order.
template <int R, int SIGN>
struct Dft;
template <int SIGN>
struct Dft<2, SIGN> {
    __device__ __forceinline__ static void run(float2 *v) {
        float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};
template <int SIGN>
struct Dft<3, SIGN> {
    __device__ __forceinline__ static void run(float2 *v) {
        // y0 = a + b + c,  y1,2 = a - (b + c)/2 +- SIGN i (sqrt(3)/2) (b - c)
        const float2 a = v[0], s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
        const float h = 0.86602540378443864676f;
        const float2 m = make_float2(a.x - 0.5f * s.x, a.y - 0.5f * s.y);
        const float2 r = mul_i<SIGN>(make_float2(h * d.x, h * d.y));
        v[0] = cadd(a, s);
        v[1] = cadd(m, r);
        v[2] = csub(m, r);
    }
};
template <int SIGN>
struct Dft<4, SIGN> {
    __device__ __forceinline__ static void run(float2 *v) {
        float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        float2 t2 = cadd(v[1], v[3]), t3 = mul_i<SIGN>(csub(v[1], v[3]));
        v[0] = cadd(t0, t2);
        v[2] = csub(t0, t2);
        v[1] = cadd(t1, t3);
        v[3] = csub(t1, t3);
    }
};
template <int SIGN>
struct Dft<8, SIGN> {
    __device__ __forceinline__ static void run(float2 *v) {
        float2 e[4] = {v[0], v[2], v[4], v[6]};
        float2 o[4] = {v[1], v[3], v[5], v[7]};
        Dft<4, SIGN>::run(e);
        Dft<4, SIGN>::run(o);
        const float h = 0.70710678118654752440f;
        // W8^1 = (1 + SIGN i)/sqrt2, W8^2 = SIGN i, W8^3 = (-1 + SIGN i)/sqrt2
        float2 o1 = SIGN > 0 ? make_float2(h * (o[1].x - o[1].y), h * (o[1].x + o[1].y))
                             : make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
        float2 o2 = mul_i<SIGN>(o[2]);
        float2 o3 = SIGN > 0 ? make_float2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y))
                             : make_float2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
        v[0] = cadd(e[0], o[0]);
        v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o1);
        v[5] = csub(e[1], o1);
        v[2] = cadd(e[2], o2);
        v[6] = csub(e[2], o2);
        v[3] = cadd(e[3], o3);
        v[7] = csub(e[3], o3);
    }
};

// ------------------------------------------------------------------ Stockham stages in LDS
template <int SIGN>
struct Dft<16, SIGN> {
    __device__ __forceinline__ static void run(float2 *v) {
        float2 e[8], o[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            e[i] = v[2 * i];
            o[i] = v[2 * i + 1];
        }
        Dft<8, SIGN>::run(e);
        Dft<8, SIGN>::run(o);
        // W16^k = exp(SIGN 2 pi i k / 16), k = 0..7
        const float c = 0.92387953251128675613f, s = 0.38268343236508977173f;
        const float h = 0.70710678118654752440f;
        const float wr[8] = {1.f, c, h, s, 0.f, -s, -h, -c};
        const float wi[8] = {0.f, s, h, c, 1.f, c, h, s};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float2 w = make_float2(wr[k], SIGN > 0 ? wi[k] : -wi[k]);
            const float2 t = (k == 0) ? o[0] : cmul(o[k], w);
            v[k] = cadd(e[k], t);
            v[k + 8] = csub(e[k], t);
        }
    }
};

template <int SIGN>
struct Dft<32, SIGN> {
    __device__ __forceinline__ static void run(float2 *v) {
        float2 e[16], o[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            e[i] = v[2 * i];
            o[i] = v[2 * i + 1];
        }
        Dft<16, SIGN>::run(e);
        Dft<16, SIGN>::run(o);
        // W32^k = exp(SIGN 2 pi i k / 32), k = 0..15
        const float wr[16] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f,
                              0.83146961230254523708f, 0.70710678118654752440f,
                              0.55557023301960222474f, 0.38268343236508977173f,
                              0.19509032201612826785f, 0.f, -0.19509032201612826785f,
                              -0.38268343236508977173f, -0.55557023301960222474f,
                              -0.70710678118654752440f, -0.83146961230254523708f,
                              -0.92387953251128675613f, -0.98078528040323044913f};
        const float wi[16] = {0.f, 0.19509032201612826785f, 0.38268343236508977173f,
                              0.55557023301960222474f, 0.70710678118654752440f,
                              0.83146961230254523708f, 0.92387953251128675613f,
                              0.98078528040323044913f, 1.f, 0.98078528040323044913f,
                              0.92387953251128675613f, 0.83146961230254523708f,
                              0.70710678118654752440f, 0.55557023301960222474f,
                              0.38268343236508977173f, 0.19509032201612826785f};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float2 w = make_float2(wr[k], SIGN > 0 ? wi[k] : -wi[k]);
            const float2 t = (k == 0) ? o[0] : cmul(o[k], w);
            v[k] = cadd(e[k], t);
            v[k + 16] = csub(e[k], t);
        }
    }
};

// ------------------------------------------------------------------ wave-level c2r of one z-line
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One line of H = 16 A complex points (A = 16: 512-point z-lines, A = 32: 1024-point ones).
// In: x[a] = X[16 a + b] (a < A), xh = Re X[H].  Out: x[16 r + d] = z[(b + 16 r) + A d],
// r < A / 16, d < 16.  L: this line's LDS region (A rows of 17), twH / twN:
// exp(-2 pi i t / H) and exp(-2 pi i t / 2H), t < H.
template <int A, int P = 16>
__device__ __forceinline__ void wave_c2r(float2 (&x)[A], float xh, float2 *L, const float2 *twH,
                                         const float2 *twN, int b) {
    static_assert(A % P == 0, "rows per lane");
    constexpr int H = P * A;
    // The mirror partner X[H - k] of k = P a + b lives in lane (P - b) % P, register A - 1 - a
    // (b = 0: the own lane, register (A - a) % A).  With 16 lanes per line that lane permutation is
    // row_mirror followed by row_ror:1 -- two DPP moves per register instead of an LDS write and
    // read of the whole line (LDS bytes count like global bytes inside a CU: 8 -> 4 KB of LDS
    // traffic per 2 KB line).
    constexpr bool DPP = C21X_ZW_DPP && P == 16;
    float2 part[DPP ? A : 1];
    if constexpr (DPP) {
#pragma unroll
        for (int a = 0; a < A; a++) {
            const float2 src = x[A - 1 - a];
            int px = __builtin_amdgcn_update_dpp(0, __float_as_int(src.x), 0x140, 0xf, 0xf, false);
            int py = __builtin_amdgcn_update_dpp(0, __float_as_int(src.y), 0x140, 0xf, 0xf, false);
            px = __builtin_amdgcn_update_dpp(0, px, 0x121, 0xf, 0xf, false);
            py = __builtin_amdgcn_update_dpp(0, py, 0x121, 0xf, 0xf, false);
            const float2 own = x[(A - a) % A];
            part[a] = (b == 0) ? own : make_float2(__int_as_float(px), __int_as_float(py));
        }
    } else {
#pragma unroll
        for (int a = 0; a < A; a++) L[a * (P + 1) + b] = x[a];
        wave_fence();
    }
    // Z[k] = E + i O, E = X[k] + conj(X[H-k]), O = (X[k] - conj(X[H-k])) exp(+2 pi i k / 2H)
#pragma unroll
    for (int a = 0; a < A; a++) {
        const int k = P * a + b;
        const int kp = (H - k) & (H - 1);  // k = 0 pairs with the Nyquist value below
        const float2 Xk = x[a];
        float2 B = DPP ? part[DPP ? a : 0] : L[(kp / P) * (P + 1) + (kp % P)];
        if (k == 0) B = make_float2(xh, 0.f);
        const float2 E = make_float2(Xk.x + B.x, Xk.y - B.y);
        const float2 D = make_float2(Xk.x - B.x, Xk.y + B.y);
        float2 w = twN[k];
        w.y = -w.y;
        const float2 O = cmul(D, w);
        x[a] = (k == 0) ? make_float2(Xk.x + xh, Xk.x - xh) : make_float2(E.x - O.y, E.y + O.x);
    }
    Dft<A, +1>::run(x);  // over a: Y_b[c]
    wave_fence();        // the partner reads are done before the region is overwritten
#pragma unroll
    for (int c = 0; c < A; c++) {
        float2 w = twH[c * b];
        w.y = -w.y;
        L[c * (P + 1) + b] = (c == 0) ? x[0] : cmul(x[c], w);
    }
    wave_fence();
#pragma unroll
    for (int r = 0; r < A / P; r++) {  // this lane's rows c = b + P r
#pragma unroll
        for (int bb = 0; bb < P; bb++) x[P * r + bb] = L[(b + P * r) * (P + 1) + bb];
        Dft<P, +1>::run(x + P * r);  // over b
    }
}

