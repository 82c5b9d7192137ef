"""Accuracy of the in-kernel window evaluation of pass X (fft_native.hip: weval_one), emulated in
numpy: node tables (W, W'h, W''h^2/2) on nodes h = 1/4 apart, quintic Hermite in fp32, against
the windows of filtering.c:80-104,357-361 evaluated in long double.  Prints the worst error in
units of the window's envelope (min(1, 3 / x^2) for the top-hat).  CPU only; diagnostic."""
import numpy as np

ld = np.longdouble


def tophat(x):
    x = np.asarray(x, ld)
    small = x < 1e-2
    xs = np.where(small, 1, x)
    w = 3.0 * (np.sin(xs) - xs * np.cos(xs)) / xs**3
    return np.where(small, 1 - x * x / 10 + x**4 / 280 - x**6 / 15120 + x**8 / 1330560, w)


def expmfp(x, ratio):
    x = np.asarray(x, ld)
    ratio = ld(ratio)
    e = np.exp(-1 / ratio)
    r2, r3 = ratio**2, ratio**3
    ts0 = 6 * r3 - e * (6 * r3 + 6 * r2 + 3 * ratio)
    ts2 = e * (2 * r2 + 0.5 * ratio) - 2 * ts0 * r2
    xs = np.where(x < 1e-3, 1, x)
    f = (xs * xs * r2 + 2 * ratio + 1) * ratio * np.cos(xs)
    f = f + (xs * xs * (r2 - r3) + ratio + 1) * np.sin(xs) / xs
    f = f * e - 2 * r2
    d = xs * ratio * xs * ratio + 1
    return np.where(x < 1e-3, ts0 + ts2 * x * x, f * (-3 * ratio / (d * d)))


def node_tables(f, h, nmax, dtype=np.float64):
    """What window_nodes_kernel computes (8th-order central differences, step h/16, in `dtype`)."""
    x = np.arange(nmax + 1, dtype=dtype) * dtype(h)
    s = dtype(h) / 16
    F = lambda y: f(np.abs(y)).astype(dtype)  # noqa: E731
    c1 = [4 / 5, -1 / 5, 4 / 105, -1 / 280]
    c2 = [8 / 5, -1 / 5, 8 / 315, -1 / 560]
    f0 = F(x)
    d1 = sum(dtype(c1[k - 1]) * (F(x + k * s) - F(x - k * s)) for k in range(1, 5)) / s
    d2 = (dtype(-205 / 72) * f0 + sum(dtype(c2[k - 1]) * (F(x + k * s) + F(x - k * s)) for k in range(1, 5))) / (s * s)
    return f0, d1 * dtype(h), d2 * dtype(h) * dtype(h) / 2


def interp32(tab, x0, h):
    f, g, q = [t.astype(np.float32) for t in tab]
    f32 = np.float32
    u = x0.astype(f32) * f32(1 / h)
    n = np.floor(u).astype(np.int64)
    t = (u - n.astype(f32)).astype(f32)
    f0, g0, q0, f1, g1, q1 = f[n], g[n], q[n], f[n + 1], g[n + 1], q[n + 1]
    A, B, C = ((f1 - f0) - g0) - q0, (g1 - g0) - 2 * q0, q1 - q0
    a3, a4, a5 = 10 * A - 4 * B + C, -15 * A + 7 * B - 2 * C, 6 * A - 3 * B + C
    p = a5
    for c in (a4, a3, q0, g0, f0):
        p = (p * t + c).astype(f32)
    return p


if __name__ == "__main__":
    h, nmax = 0.25, int(200 / 0.25) + 2
    rng = np.random.default_rng(1)
    x0 = np.concatenate([rng.uniform(0, 4, 200000), rng.uniform(0, 190, 800000)]).astype(np.float32)
    env = np.minimum(1.0, 3 / np.maximum(x0.astype(np.float64), 1e-9) ** 2)
    tab = node_tables(tophat, h, nmax)
    err = np.abs(interp32(tab, x0, h).astype(np.float64) - tophat(x0.astype(ld)).astype(np.float64)) / env
    print(f"top-hat: max |err| / envelope {err.max():.3e}, rms {np.sqrt((err**2).mean()):.3e}")
    for ratio in (25.483 / 0.93, 25.483 / 10, 25.483 / 38.0, 0.2):
        tab = node_tables(lambda x: expmfp(x, ratio), h, nmax)  # noqa: B023
        ex = expmfp(x0.astype(ld), ratio).astype(np.float64)
        e = np.abs(interp32(tab, x0, h).astype(np.float64) - ex)
        print(f"exp-MFP ratio {ratio:7.3f}: max |err| {e.max():.3e} (max |W| {np.abs(ex).max():.3f})")


# ---- direct fp32 evaluation beyond the node tables (weval_one, x >= X_SWITCH) ----------------
def sincos32(x):
    """fp32 sin / cos as the kernel forms them: n = rint(x 2/pi), two-constant Cody-Waite reduction
    with FMA (emulated in float64 then rounded: an FMA rounds once), cephes minimax kernels."""
    f32 = np.float32
    x = x.astype(f32)
    n = np.rint(x * f32(0.6366197723675814)).astype(f32)
    hi, lo = f32(1.5707963705062866), f32(-4.371138828673793e-08)
    r = (x.astype(np.float64) - n.astype(np.float64) * np.float64(hi)).astype(f32)  # fma
    r = (r.astype(np.float64) - n.astype(np.float64) * np.float64(lo)).astype(f32)  # fma
    z = (r * r).astype(f32)
    s = ((f32(-1.9515295891e-4) * z + f32(8.3321608736e-3)) * z + f32(-1.6666654611e-1)).astype(f32)
    s = (s * z * r + r).astype(f32)
    c = ((f32(2.443315711809948e-5) * z + f32(-1.388731625493765e-3)) * z + f32(4.166664568298827e-2)).astype(f32)
    c = (c * z * z - f32(0.5) * z + f32(1.0)).astype(f32)
    q = n.astype(np.int64) & 3
    sn = np.where(q == 0, s, np.where(q == 1, c, np.where(q == 2, -s, -c))).astype(f32)
    cs = np.where(q == 0, c, np.where(q == 1, -s, np.where(q == 2, -c, s))).astype(f32)
    return sn, cs


def tophat_direct32(x0):
    f32 = np.float32
    x = x0.astype(f32)
    sn, cs = sincos32(x)
    inv = (f32(1) / x).astype(f32)
    return (f32(3) * (sn - x * cs).astype(f32) * (inv * inv).astype(f32) * inv).astype(f32)


def expmfp_direct32(x0, d, ratio):
    """x = x0 + d (d: the rounding residual of kR); constants rounded to float."""
    f32 = np.float32
    ratio64 = np.float64(ratio)
    e = np.exp(-1 / ratio64)
    r, r2, r3, et = f32(ratio64), f32(ratio64**2), f32(ratio64**3), f32(e)
    x = x0.astype(f32)
    s0, c0 = sincos32(x)
    sn = (s0 + d * c0).astype(f32)
    cs = (c0 - d * s0).astype(f32)
    xx = (x * x + f32(2) * x * d).astype(f32)  # (x0 + d)^2 to first order
    f = ((xx * r2 + (f32(2) * r + f32(1))) * r * cs).astype(f32)
    f = (f + (xx * (r2 - r3) + (r + f32(1))) * sn / (x + d)).astype(f32)
    f = (f * et - f32(2) * r2).astype(f32)
    dd = (xx * r2 + f32(1)).astype(f32)
    return (f * (f32(-3) * r / (dd * dd))).astype(f32)


def check_direct():
    rng = np.random.default_rng(2)
    x0 = rng.uniform(12, 2000, 1000000).astype(np.float32)
    env = 3 / x0.astype(np.float64) ** 2
    err = np.abs(tophat_direct32(x0).astype(np.float64) - tophat(x0.astype(ld)).astype(np.float64)) / env
    print(f"top-hat direct fp32, x in [12, 2000]: max |err| / envelope {err.max():.3e}, rms {np.sqrt((err**2).mean()):.3e}")
    for ratio in (27.4, 2.548, 0.671, 0.05):
        xd = rng.uniform(12, 2000, 400000)
        x0 = xd.astype(np.float32)
        d = (xd - x0.astype(np.float64)).astype(np.float32)
        ex = expmfp(xd.astype(ld), ratio).astype(np.float64)
        got = expmfp_direct32(x0, d, ratio).astype(np.float64)
        # envelope: the running maximum of |W| towards larger x
        o = np.argsort(xd)
        envr = np.maximum.accumulate(np.abs(ex[o])[::-1])[::-1]
        e = np.abs(got - ex)[o] / envr
        print(f"exp-MFP direct fp32 ratio {ratio:6.3f}: max |err| / envelope {e.max():.3e}, rms {np.sqrt((e**2).mean()):.3e}")


if __name__ == "__main__":
    check_direct()
