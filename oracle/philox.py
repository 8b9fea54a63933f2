"""CPU restatement of the device normal generator (TEST INFRASTRUCTURE ONLY).

Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
Random123 reference constants) followed by Box-Muller, exactly as csrc/nsdiff_chain.hip
does it: counter = (chain_lo, chain_hi, k, offset_lo), key = (seed_lo, seed_hi + offset_hi).
The reference itself uses torch.randn_like (nsdiff_utils.py:130,273); this generator is
the build's on-device replacement and is validated statistically plus bit-for-bit on the
integer part against the published known-answer vectors.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays; returns four uint32 arrays."""
    c = [np.asarray(v, dtype=np.uint64) & MASK for v in (c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def normals(seed: int, offset: int, n_chains: int, steps: int, nf: int) -> np.ndarray:
    """[n_chains, steps, nf] float32 normals = what mpv_nsdiff_philox_normals writes.

    Draw k, feature f takes normal (k*nf+f)&3 of Philox block (k*nf)>>2 when nf is 1 or 2 (all four
    normals of a block are used), normal f of block k otherwise (csrc/chain_common.h::step_normals)."""
    packed = nf in (1, 2)
    n_blocks = (steps * nf + 3) // 4 if packed else steps
    chain = np.repeat(np.arange(n_chains, dtype=np.uint64), n_blocks)
    blk = np.tile(np.arange(n_blocks, dtype=np.uint64), n_chains)
    seed_lo, seed_hi = seed & 0xFFFFFFFF, ((seed >> 32) + (offset >> 32)) & 0xFFFFFFFF
    x = philox4x32_10(chain & MASK, chain >> np.uint64(32), blk, np.full_like(blk, offset & 0xFFFFFFFF), seed_lo, seed_hi)
    inv24 = np.float32(2.0 ** -24)
    z = np.empty((n_chains * n_blocks, 4), dtype=np.float32)
    for i in range(2):
        u1 = ((x[2 * i] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * inv24
        u2 = (x[2 * i + 1] >> np.uint32(8)).astype(np.float32) * inv24
        rad = np.sqrt(np.float32(-2.0) * np.log(u1.astype(np.float64))).astype(np.float32)
        ang = 2.0 * np.pi * u2.astype(np.float64)
        z[:, 2 * i] = rad * np.cos(ang).astype(np.float32)
        z[:, 2 * i + 1] = rad * np.sin(ang).astype(np.float32)
    if packed:
        return z.reshape(n_chains, n_blocks * 4)[:, : steps * nf].reshape(n_chains, steps, nf)
    return z[:, :nf].reshape(n_chains, steps, nf)
