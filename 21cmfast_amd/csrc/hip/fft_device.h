// fft_device.h -- device-side building blocks of the transform kernels of fft_native.hip (and of experimental
// kernels kept outside the product tree): complex helpers, the small DFTs in registers and the
// wave-level complex-to-real line transform.  Included INSIDE each translation unit's anonymous
// namespace; the arithmetic of a line is the same instruction sequence wherever it is instantiated
// (-ffp-contract=off), which is what keeps the fused and the unfused passes bit-identical.
#pragma once
#ifndef C21X_ZW_DPP
#define C21X_ZW_DPP 0
#endif

// ------------------------------------------------------------------ complex helpers
// Round 5: what the SLP vectoriser made of this scalar arithmetic cost the transforms a quarter of their
// vector instructions -- it packs pairs of additions / multiplications into v_pk_add_f32 / v_pk_mul_f32 and
// builds the operand pairs with v_mov (four moves per complex product in the c2r pre-processing; 300 of
// the fused pass Z's 2050 vector instructions were moves).  Two ways out, both bit-identical to the old
// code (same multiplications, additions and roundings; -ffp-contract=off), both measured:
//   C21X_PKASM = 1: the complex products and the multiplications by +-i hand-written as the packed
//     instructions they are (op_sel and neg modifiers: a complex product is three instructions, no move):
//     fused pass Z 2047 -> 1496 vector instructions, 0.309 -> 0.27 ms per 512^3 radius;
//   the default build: NO packing at all (-fno-slp-vectorize, Makefile): 1976 vector instructions, all
//     plain -- and as fast (0.26-0.27 ms), 3 % faster at 1024^3, because a packed fp32 instruction issues in
//     ~8 cycles per wave on gfx950 where a plain one takes ~2.7 (tools/valu_rate_probe.hip).  No inline
//     assembly in the default build.
#ifndef C21X_PKASM
#define C21X_PKASM 0  // measured equal to the scalar forms built with -fno-slp-vectorize (the default build)
#endif
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
#if C21X_PKASM
// a b = (a.x b.x - a.y b.y, a.y b.x + a.x b.y):  t = (a.x, a.y) b.x,  u = (a.y, a.x) b.y,  t -+ u
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    float2 t, u, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(u) : "v"(a), "v"(b));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(t), "v"(u));
    return r;
}
// a conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y): bitwise cmul(a, (b.x, -b.y))
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    float2 t, u, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(u) : "v"(a), "v"(b));
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(t), "v"(u));
    return r;
}
// a + i b = (a.x - b.y, a.y + b.x),  a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ float2 cadd_i(float2 a, float2 b) {
    float2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float2 csub_i(float2 a, float2 b) {
    float2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a.x + b.x, a.y - b.y) and (a.x - b.x, a.y + b.y): X + conj(B), X - conj(B)
__device__ __forceinline__ float2 cadd_c(float2 a, float2 b) {
    float2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float2 csub_c(float2 a, float2 b) {
    float2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// h a, -h a for a real h
__device__ __forceinline__ float2 cscale(float h, float2 a) {
    float2 r, hh = make_float2(h, h);
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(hh), "v"(a));
    return r;
}
__device__ __forceinline__ float2 cscale_neg(float h, float2 a) {
    float2 r, hh = make_float2(h, h);
    asm("v_pk_mul_f32 %0, %1, %2 neg_lo:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(hh), "v"(a));
    return r;
}
#else
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return cmul(a, make_float2(b.x, -b.y)); }
__device__ __forceinline__ float2 cadd_i(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }
__device__ __forceinline__ float2 csub_i(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }
__device__ __forceinline__ float2 cadd_c(float2 a, float2 b) { return make_float2(a.x + b.x, a.y - b.y); }
__device__ __forceinline__ float2 csub_c(float2 a, float2 b) { return make_float2(a.x - b.x, a.y + b.y); }
__device__ __forceinline__ float2 cscale(float h, float2 a) { return make_float2(h * a.x, h * a.y); }
__device__ __forceinline__ float2 cscale_neg(float h, float2 a) { return make_float2(-h * a.x, -h * a.y); }
#endif
// multiply by +i (SIGN > 0) or -i (SIGN < 0)
template <int SIGN>
__device__ __forceinline__ float2 mul_i(float2 a) {
    return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// a + SIGN i b, a - SIGN i b
template <int SIGN>
__device__ __forceinline__ float2 cadd_si(float2 a, float2 b) { return SIGN > 0 ? cadd_i(a, b) : csub_i(a, b); }
template <int SIGN>
__device__ __forceinline__ float2 csub_si(float2 a, float2 b) { return SIGN > 0 ? csub_i(a, b) : cadd_i(a, b); }

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
        // (t1 +- SIGN i d: the same additions as t1 +- mul_i(d), x + (-y) = x - y bit for bit)
        float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        float2 t2 = cadd(v[1], v[3]), d = csub(v[1], v[3]);
        v[0] = cadd(t0, t2);
        v[2] = csub(t0, t2);
        v[1] = cadd_si<SIGN>(t1, d);
        v[3] = csub_si<SIGN>(t1, d);
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
        //   SIGN > 0: o1 = h (x - y, x + y) = h (o + i o),  o3 = (-h (x + y), h (x - y)) = -h (o - i o)
        //   SIGN < 0: o1 = h (x + y, y - x) = h (o - i o),  o3 = (h (y - x), -h (x + y)) = -h (o + i o)
        // (y - x = -(x - y) and (-h) s = -(h s) bit for bit)
        const float2 op = cadd_i(o[1], o[1]), om = csub_i(o[1], o[1]);
        const float2 pp = cadd_i(o[3], o[3]), pm = csub_i(o[3], o[3]);
        const float2 o1 = SIGN > 0 ? cscale(h, op) : cscale(h, om);
        const float2 o3 = SIGN > 0 ? cscale_neg(h, pm) : cscale_neg(h, pp);
        v[0] = cadd(e[0], o[0]);
        v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o1);
        v[5] = csub(e[1], o1);
        v[2] = cadd_si<SIGN>(e[2], o[2]);
        v[6] = csub_si<SIGN>(e[2], o[2]);
        v[3] = cadd(e[3], o3);
        v[7] = csub(e[3], o3);
    }
};

// ------------------------------------------------------------------ constant twiddles
// a w for w = (sx W[IX], sy W[IY]) taken out of ONE register pair W = (cos, sin): the twiddles of the 16-
// and 32-point DFTs are (c, s), (s, c), (-s, c), (-c, s) of three angles, so three pairs (and h) serve
// them all through op_sel / neg modifiers.  NX / NY: negate the real / imaginary part of w.
#if C21X_PKASM
template <int IX, int NX, int IY, int NY>
__device__ __forceinline__ float2 cmul_k(float2 a, float2 W) {
    float2 t, u, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,%3] op_sel_hi:[1,%3] neg_lo:[0,%4] neg_hi:[0,%4]"
        : "=v"(t) : "v"(a), "v"(W), "n"(IX), "n"(NX));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,%3] op_sel_hi:[0,%3] neg_lo:[0,%4] neg_hi:[0,%4]"
        : "=v"(u) : "v"(a), "v"(W), "n"(IY), "n"(NY));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(t), "v"(u));
    return r;
}
// (sa t.x + sb t.y, sc t.x + sd t.y) in one instruction (N* = 1: minus)
template <int NA, int NB, int NC, int ND>
__device__ __forceinline__ float2 cmix(float2 t) {
    float2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[%2,%3] neg_hi:[%4,%5]"
        : "=v"(r) : "v"(t), "n"(NA), "n"(NB), "n"(NC), "n"(ND));
    return r;
}
#else
template <int IX, int NX, int IY, int NY>
__device__ __forceinline__ float2 cmul_k(float2 a, float2 W) {
    const float wx = IX ? W.y : W.x, wy = IY ? W.y : W.x;
    return cmul(a, make_float2(NX ? -wx : wx, NY ? -wy : wy));
}
template <int NA, int NB, int NC, int ND>
__device__ __forceinline__ float2 cmix(float2 t) {
    return make_float2((NA ? -t.x : t.x) + (NB ? -t.y : t.y), (NC ? -t.x : t.x) + (ND ? -t.y : t.y));
}
#endif
// a (h + SIGN i h) and a (-h + SIGN i h) from t = h a:
//   (h, h): (t.x - t.y, t.x + t.y);  (h, -h): (t.x + t.y, t.y - t.x);
//   (-h, h): (-t.x - t.y, t.x - t.y);  (-h, -h): (t.y - t.x, -t.x - t.y)
template <int SIGN>
__device__ __forceinline__ float2 cmul_hh(float2 a, float h) {
    const float2 t = cscale(h, a);
    return SIGN > 0 ? cadd_i(t, t) : csub_i(t, t);
}
template <int SIGN>
__device__ __forceinline__ float2 cmul_mhh(float2 a, float h) {
    const float2 t = cscale(h, a);
    return SIGN > 0 ? cmix<1, 1, 0, 1>(t) : cmix<1, 0, 1, 1>(t);
}

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
        // W16^k = exp(SIGN 2 pi i k / 16), k = 0..7:
        //   (1, 0), (c, s), (h, h), (s, c), (0, 1), (-s, c), (-h, h), (-c, s) with the imaginary parts times SIGN
        const float2 W = make_float2(0.92387953251128675613f, 0.38268343236508977173f);
        const float h = 0.70710678118654752440f;
        constexpr int NY = SIGN > 0 ? 0 : 1;
        float2 t[8];
        t[0] = o[0];
        t[1] = cmul_k<0, 0, 1, NY>(o[1], W);
        t[2] = cmul_hh<SIGN>(o[2], h);
        t[3] = cmul_k<1, 0, 0, NY>(o[3], W);
        t[5] = cmul_k<1, 1, 0, NY>(o[5], W);
        t[6] = cmul_mhh<SIGN>(o[6], h);
        t[7] = cmul_k<0, 1, 1, NY>(o[7], W);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k == 4) {  // W = SIGN i (multiplications by 0 and 1 dropped: the sign of a zero at most)
                v[4] = cadd_si<SIGN>(e[4], o[4]);
                v[12] = csub_si<SIGN>(e[4], o[4]);
                continue;
            }
            v[k] = cadd(e[k], t[k]);
            v[k + 8] = csub(e[k], t[k]);
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
        // W32^k = exp(SIGN 2 pi i k / 32), k = 0..15, out of three (cos, sin) pairs and h
        const float2 W1 = make_float2(0.98078528040323044913f, 0.19509032201612826785f);
        const float2 W2 = make_float2(0.92387953251128675613f, 0.38268343236508977173f);
        const float2 W3 = make_float2(0.83146961230254523708f, 0.55557023301960222474f);
        const float h = 0.70710678118654752440f;
        constexpr int NY = SIGN > 0 ? 0 : 1;
        float2 t[16];
        t[0] = o[0];
        t[1] = cmul_k<0, 0, 1, NY>(o[1], W1);
        t[2] = cmul_k<0, 0, 1, NY>(o[2], W2);
        t[3] = cmul_k<0, 0, 1, NY>(o[3], W3);
        t[4] = cmul_hh<SIGN>(o[4], h);
        t[5] = cmul_k<1, 0, 0, NY>(o[5], W3);
        t[6] = cmul_k<1, 0, 0, NY>(o[6], W2);
        t[7] = cmul_k<1, 0, 0, NY>(o[7], W1);
        t[9] = cmul_k<1, 1, 0, NY>(o[9], W1);
        t[10] = cmul_k<1, 1, 0, NY>(o[10], W2);
        t[11] = cmul_k<1, 1, 0, NY>(o[11], W3);
        t[12] = cmul_mhh<SIGN>(o[12], h);
        t[13] = cmul_k<0, 1, 1, NY>(o[13], W3);
        t[14] = cmul_k<0, 1, 1, NY>(o[14], W2);
        t[15] = cmul_k<0, 1, 1, NY>(o[15], W1);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k == 8) {
                v[8] = cadd_si<SIGN>(e[8], o[8]);
                v[24] = csub_si<SIGN>(e[8], o[8]);
                continue;
            }
            v[k] = cadd(e[k], t[k]);
            v[k + 16] = csub(e[k], t[k]);
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
        const float2 E = cadd_c(Xk, B);
        const float2 D = csub_c(Xk, B);
        const float2 O = cmulc(D, twN[k]);
        x[a] = (k == 0) ? make_float2(Xk.x + xh, Xk.x - xh) : cadd_i(E, O);
    }
    Dft<A, +1>::run(x);  // over a: Y_b[c]
    wave_fence();        // the partner reads are done before the region is overwritten
#pragma unroll
    for (int c = 0; c < A; c++) {
        L[c * (P + 1) + b] = (c == 0) ? x[0] : cmulc(x[c], twH[c * b]);
    }
    wave_fence();
#pragma unroll
    for (int r = 0; r < A / P; r++) {  // this lane's rows c = b + P r
#pragma unroll
        for (int bb = 0; bb < P; bb++) x[P * r + bb] = L[(b + P * r) * (P + 1) + bb];
        Dft<P, +1>::run(x + P * r);  // over b
    }
}

