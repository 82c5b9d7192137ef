"""LDS bank-conflict model of the pass-Z tile layouts (diagnostic, CPU only).

Banking rules from /opt/skills/guides/MI355X_MICROARCH.md (LDS):
  ds_read_b64   two 32-lane groups, bank = (byte/4) mod 64
  ds_write_b64  four contiguous 16-lane groups, bank = (byte/4) mod 32
A group costs max over banks of the number of distinct dwords on that bank.
Prints LDS-array cycles per workgroup for every access pattern of z_transform / the fused
kernel's output read, for a list of candidate layouts.
"""
import sys
from collections import defaultdict

H, LZ, T = 256, 8, 256


def cost(addrs_per_lane, write):
    """addrs_per_lane: list (one per lane of a wave, None = inactive) of float2 element indices"""
    groups = [range(g * 16, g * 16 + 16) for g in range(4)] if write else [range(0, 32), range(32, 64)]
    nb = 32 if write else 64
    total = 0
    for g in groups:
        banks = defaultdict(set)
        for l in g:
            a = addrs_per_lane[l]
            if a is None:
                continue
            for d in (2 * a, 2 * a + 1):
                banks[d % nb].add(d)
        total += max((len(v) for v in banks.values()), default=0)
    return total


def waves(fn, n_items):
    """fn(item) -> element index or None; items are thread-major: item = tid + T*u"""
    cyc = ideal = 0
    for base in range(0, n_items, 64):
        lanes = [fn(base + l) if base + l < n_items else None for l in range(64)]
        yield lanes


def run(layout, name):
    addr = layout
    res = {}

    def acc(key, lanes, write):
        c = cost(lanes, write)
        active = sum(1 for a in lanes if a is not None)
        ideal = (4 if write else 2) if active else 0
        r = res.setdefault(key, [0, 0])
        r[0] += c
        r[1] += ideal

    # 1. fill
    nf4 = LZ * H // 2
    for which in (0, 1):
        for lanes in waves(lambda f: addr((2 * f) % H + which, (2 * f) // H), nf4):
            acc("fill write", lanes, True)
    # 2. pre-processing (k = 1 .. H/2, pairs k and H-k; k = 0 handled by LZ lanes, ignored)
    npre = (H // 2 + 1) * LZ
    for lanes in waves(lambda i: addr(i // LZ, i % LZ) if i // LZ > 0 else None, npre):
        acc("pre read A", lanes, False)
        acc("pre write A", lanes, True)
    for lanes in waves(lambda i: addr(H - i // LZ, i % LZ) if i // LZ > 0 else None, npre):
        acc("pre read B", lanes, False)
        acc("pre write B", lanes, True)
    # 3. FFT stages
    radices, log2s = [], 0
    L = H.bit_length() - 1
    radices = [8] * (L // 3) + ({0: [], 1: [2], 2: [4]}[L % 3])
    for R in radices:
        NB = H // R
        items = NB * LZ
        for k in range(R):
            for lanes in waves(lambda i: addr(i // LZ + k * NB, i % LZ), items):
                acc(f"fft r{R} s{1 << log2s} read", lanes, False)
        for j in range(R):
            def w(i):
                col, b = i % LZ, i // LZ
                p, q = b >> log2s, b & ((1 << log2s) - 1)
                return addr(q + ((R * p) << log2s) + (j << log2s), col)
            for lanes in waves(w, items):
                acc(f"fft r{R} s{1 << log2s} write", lanes, True)
        log2s += R.bit_length() - 1
    # 4. output read
    for lanes in waves(lambda f: addr(f % H, f // H), LZ * H):
        acc("output read", lanes, False)
    tot = sum(v[0] for v in res.values())
    ideal = sum(v[1] for v in res.values())
    print(f"== {name}: {tot} cycles per workgroup and grid (conflict-free: {ideal}, x{tot / ideal:.2f})")
    if "-v" in sys.argv:
        for k, v in res.items():
            print(f"     {k:24s} {v[0]:6d}  (ideal {v[1]})")
    return tot


def linear(zrow):
    return lambda row, col: row * zrow + col


def xor_swizzle(row, col):
    # rows of LZ = 8 float2 (no padding); 16-dword unit index and column both swizzled by the row
    q = (row ^ (row >> 2) ^ (row >> 4) ^ (row >> 6)) & 3
    unit = (row & ~3) | q
    c = col ^ ((row >> 2) & 7)
    return unit * 8 + c


if __name__ == "__main__":
    for z in (8, 9, 10, 11, 12, 13, 16, 17):
        run(linear(z), f"linear ZROW={z}")
    run(xor_swizzle, "xor swizzle, ZROW=8")
