"""Test helper: numpy restatement of the noise generator of csrc/lbbnn_device.h (Philox4x32-10 of Salmon et al., "Parallel
random numbers: as easy as 1, 2, 3", SC'11, keyed and counted as philox_normal4 does it, then Box-Muller on two pairs).
Not part of the product and not a restatement of the reference (which draws with torch.randn): it pins WHAT the kernels
draw -- the standard generator, checked against Random123's known-answer vectors -- so that a change of the device code that
alters the stream (another round count, another keying) cannot pass as "still normal-looking"."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint64 arrays holding 32-bit values; returns four uint64 arrays (32-bit values)."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(W0)) & np.uint64(MASK)
        k1 = (k1 + np.uint64(W1)) & np.uint64(MASK)
    return c0, c1, c2, c3


def key_of(seed, offset):
    """philox_normal4's key: the RNG state {seed, offset} enters through the key, the element index through the counter."""
    seed, offset = int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1)
    k0 = (seed & MASK) ^ (((offset & MASK) * 0x9E3779B9) & MASK)
    k1 = ((seed >> 32) & MASK) ^ ((offset >> 32) & MASK) ^ ((offset & MASK) >> 7)
    return k0, k1


def normal4(seed, offset, stream, ctr0, ctr1):
    """Four N(0,1) per counter, as philox_normal4: u = (x + 1) 2^-32 for the radius, x 2^-32 for the angle (fp32 roundings
    of the uint32 -> float conversions included), (r cos, r sin) per pair.  float64 arithmetic after that: the device uses the
    ~1 ulp hardware log2 / sqrt / sin / cos, so compare with a tolerance of a few 1e-6."""
    ctr0 = np.asarray(ctr0, dtype=np.uint64)
    ctr1 = np.asarray(ctr1, dtype=np.uint64)
    ctr0, ctr1 = np.broadcast_arrays(ctr0, ctr1)
    k0, k1 = key_of(seed, offset)
    x, y, z, w = philox4x32_10(ctr0 & np.uint64(MASK), ctr0 >> np.uint64(32), ctr1, np.full(ctr0.shape, stream, np.uint64), k0, k1)
    f = lambda v: v.astype(np.float32)                                      # v_cvt_f32_u32: round to nearest even
    u0 = ((f(x) + np.float32(1.0)) * np.float32(2.3283064365386963e-10)).astype(np.float64)
    u1 = (f(y) * np.float32(2.3283064365386963e-10)).astype(np.float64)
    u2 = ((f(z) + np.float32(1.0)) * np.float32(2.3283064365386963e-10)).astype(np.float64)
    u3 = (f(w) * np.float32(2.3283064365386963e-10)).astype(np.float64)
    ra, rb = np.sqrt(-2.0 * np.log(u0)), np.sqrt(-2.0 * np.log(u2))
    return np.stack([ra * np.cos(2 * np.pi * u1), ra * np.sin(2 * np.pi * u1),
                     rb * np.cos(2 * np.pi * u3), rb * np.sin(2 * np.pi * u3)], axis=-1)


def normal_matrix(seed, offset, stream, rows, cols, row_base=0):
    """The (rows, cols) draw of lbbnn_philox_normal / the GEMM epilogues: element (r, c) = counter (row_base + r, c // 4)[c % 4]."""
    g = (cols + 3) // 4
    r = np.arange(rows, dtype=np.uint64)[:, None] + np.uint64(row_base)
    cg = np.arange(g, dtype=np.uint64)[None, :]
    return normal4(seed, offset, stream, r, cg).reshape(rows, 4 * g)[:, :cols]


def normal_vector(seed, offset, stream, n):
    """The 1-D draw of the flow kernels: element i = counter (i // 4, 0)[i % 4]."""
    g = (n + 3) // 4
    return normal4(seed, offset, stream, np.arange(g, dtype=np.uint64), np.zeros(g, dtype=np.uint64)).reshape(-1)[:n]


# Random123 kat_vectors, philox4x32 with 10 rounds: (counter[4], key[2]) -> output[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]
