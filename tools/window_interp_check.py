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
