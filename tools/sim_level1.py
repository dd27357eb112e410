"""Level-1 survivor rates of K1b for other table geometries (CPU simulation of the kernel's test on the
cfg2 set and 16 MiB of the T / U haystacks; the 2^14 x {X32, Y32} row reproduces the measured 0.5 %)."""
import sys
sys.path.insert(0, "/root/repo/tests")
import numpy as np, gen
pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
n = 16 << 20
for name, hay in (("T", gen.gen_textlike(n, 11, pats)), ("U", gen.gen_uniform(n, gen.AZ, 12))):
    h = np.frombuffer(hay, dtype=np.uint8).astype(np.uint32)
    def gram(a, off):  # 4-byte little-endian gram at positions a + off
        return h[off:off + len(a)] | (h[off + 1:off + 1 + len(a)] << 8) | (h[off + 2:off + 2 + len(a)] << 16) | (h[off + 3:off + 3 + len(a)] << 24)
    P = np.array([list(p[:5]) for p in pats], dtype=np.uint32)
    wx = P[:, 1] | (P[:, 2] << 8) | (P[:, 3] << 16) | (P[:, 4] << 24)
    wy = P[:, 0] | (P[:, 1] << 8) | (P[:, 2] << 16) | (P[:, 3] << 24)
    def H(W):
        return (((W & 0xFFFFFF).astype(np.uint64) * 0x9E3779 + W.astype(np.uint64)) & 0xFFFFFFFF).astype(np.uint32)
    for lg in (15, 14, 13, 12):
        X = np.zeros(1 << lg, dtype=np.uint32); Y = np.zeros(1 << lg, dtype=np.uint32)
        ex, ey = H(wx) >> (32 - lg), H(wy) >> (32 - lg)
        np.bitwise_or.at(X, ex, (1 << (P[:, 0] & 31)).astype(np.uint32) | (1 << (wx & 31)).astype(np.uint32))
        np.bitwise_or.at(Y, ey, (1 << (P[:, 4] & 31)).astype(np.uint32))
        np.bitwise_or.at(X, ey, (1 << (wy & 31)).astype(np.uint32))
        m = (n - 8) // 2
        j = np.arange(0, 2 * m, 2)
        W = h[j + 1] | (h[j + 2] << 8) | (h[j + 3] << 16) | (h[j + 4] << 24)
        e = H(W) >> (32 - lg)
        gate = (X[e] >> (W & 31)) & 1
        p0 = gate & ((X[e] >> (h[j] & 31)) & 1)
        p1 = gate & ((Y[e] >> (h[j + 5] & 31)) & 1)
        surv = (p0.sum() + p1.sum()) / (2 * m)
        print(f"{name} table 2^{lg}: survivors {100 * surv:.3f} %  (X density {np.unpackbits(X.view(np.uint8)).mean():.4f})")

print("---- 2^15 entries of 32 bits: X16 | Y16 << 16")
for name, hay in (("T", gen.gen_textlike(n, 11, pats)), ("U", gen.gen_uniform(n, gen.AZ, 12))):
    h = np.frombuffer(hay, dtype=np.uint8).astype(np.uint32)
    P = np.array([list(p[:5]) for p in pats], dtype=np.uint32)
    wx = P[:, 1] | (P[:, 2] << 8) | (P[:, 3] << 16) | (P[:, 4] << 24)
    wy = P[:, 0] | (P[:, 1] << 8) | (P[:, 2] << 16) | (P[:, 3] << 24)
    m = (n - 8) // 2
    j = np.arange(0, 2 * m, 2)
    W = h[j + 1] | (h[j + 2] << 8) | (h[j + 3] << 16) | (h[j + 4] << 24)
    for lg, nb, gate_mode in ((15, 16, "x"), (15, 16, "none"), (15, 16, "hash"), (16, 8, "none")):
        X = np.zeros(1 << lg, dtype=np.uint32); Y = np.zeros(1 << lg, dtype=np.uint32)
        ex, ey = H(wx) >> (32 - lg), H(wy) >> (32 - lg)
        def gbit(Wv):  # which bit of X doubles as the gate
            if gate_mode == "x": return Wv % nb
            if gate_mode == "hash": return (H(Wv) >> 7) % nb
            return None
        bx = (1 << (P[:, 0] % nb)).astype(np.uint32)
        if gate_mode != "none": bx = bx | (1 << gbit(wx)).astype(np.uint32)
        np.bitwise_or.at(X, ex, bx)
        np.bitwise_or.at(Y, ey, (1 << (P[:, 4] % nb)).astype(np.uint32))
        if gate_mode != "none": np.bitwise_or.at(X, ey, (1 << gbit(wy)).astype(np.uint32))
        e = H(W) >> (32 - lg)
        gate = 1 if gate_mode == "none" else (X[e] >> gbit(W)) & 1
        p0 = gate & ((X[e] >> (h[j] % nb)) & 1)
        p1 = gate & ((Y[e] >> (h[j + 5] % nb)) & 1)
        print(f"{name} 2^{lg} entries, {nb}-bit masks, gate {gate_mode}: survivors {100 * (p0.sum() + p1.sum()) / (2 * m):.3f} %")
