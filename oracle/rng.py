"""NumPy restatement of the device random draws of mici_amd/csrc/k_rng.hip (test infrastructure only).

The reference draws from a NumPy Generator on the host (transitions.py:136-142, 300-309, 383-386; systems.py:365-366);
the device path replaces the SOURCE of the draws, not what is done with them, by a counter-based generator:
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 - constants and
round function as published; checked below against the paper's known-answer vectors) keyed by a 64-bit seed, counter =
(chain index 64 bits | transition 40 bits | purpose 2 bits | block 22 bits).  Integer arithmetic: bit-exact parity."""

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
PURPOSE_NORMAL, PURPOSE_UNIFORM, PURPOSE_STEPS = 0, 1, 2


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over arrays of uint32 counters; returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32).copy() for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def draw_block(seed, chain, transition, purpose, block):
    seed, transition = int(seed), int(transition)
    chain = np.asarray(chain, dtype=np.uint64)
    block = np.asarray(block, dtype=np.uint64)
    c3 = (np.uint64((transition >> 32) & 0xFF) | np.uint64((purpose & 3) << 8) | ((block & np.uint64(0x3FFFFF)) << np.uint64(10)))
    return philox4x32_10((chain & np.uint64(0xFFFFFFFF)).astype(np.uint32), (chain >> np.uint64(32)).astype(np.uint32),
                         np.uint32(transition & 0xFFFFFFFF), c3.astype(np.uint32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def u53(a, b):
    return ((a >> np.uint32(5)).astype(np.float64) * 67108864.0 + (b >> np.uint32(6)).astype(np.float64)) / 9007199254740992.0


def normal(seed, chain_offset, n_chains, dim, transition):
    """z[n_chains, dim] of momentum refresh number `transition`."""
    pairs = (dim + 1) // 2
    chain = (np.uint64(chain_offset) + np.arange(n_chains, dtype=np.uint64))[:, None]
    k = np.arange(pairs, dtype=np.uint64)[None, :]
    x0, x1, x2, x3 = draw_block(seed, chain, transition, PURPOSE_NORMAL, k)
    ua, ub = u53(x0, x1), u53(x2, x3)
    r = np.sqrt(-2.0 * np.log(1.0 - ua))
    th = 6.283185307179586476925286766559 * ub
    z = np.empty((n_chains, 2 * pairs))
    z[:, 0::2], z[:, 1::2] = r * np.cos(th), r * np.sin(th)
    return z[:, :dim]


def uniform(seed, chain_offset, n_chains, transition):
    chain = np.uint64(chain_offset) + np.arange(n_chains, dtype=np.uint64)
    x0, x1, _, _ = draw_block(seed, chain, transition, PURPOSE_UNIFORM, np.uint64(0))
    return u53(x0, x1)


def steps(seed, chain_offset, n_chains, transition, lo, hi):
    chain = np.uint64(chain_offset) + np.arange(n_chains, dtype=np.uint64)
    x0, _, _, _ = draw_block(seed, chain, transition, PURPOSE_STEPS, np.uint64(0))
    return (lo + ((x0.astype(np.uint64) * np.uint64(hi - lo)) >> np.uint64(32))).astype(np.int32)
