#!/usr/bin/env python3
"""Lane-accurate NumPy model of csrc/k_implicit_blk16.hip (the c4 kernel: D <= 256, one 8-wave workgroup per chain).

Every index formula of the HIP kernel - tile ownership of a wave's 17 register slots, the MFMA operand /
accumulator lane layouts, the padded LDS panel layout, the in-tile 4-wide sweep that inverts the 16 x 16 pivot
block, the block-16 symmetric sweep (full inverse) and its trailing-only form (block LDL^T) with the forward /
backward substitutions - is restated here on arrays indexed [wave][slot][lane][reg] and checked against
numpy.linalg, so that layout mistakes are found on the CPU, not on the GPU box.

    python tools/sim_blk16.py          # prints max errors; exits non-zero on failure
"""

import sys

import numpy as np

NT = 16
NW = 8   # waves per chain
NS = 17  # tiles per wave
CS = 18  # LDS panel: doubles per column (16 + 2 padding)

LANE = np.arange(64)
G = LANE >> 4
J = LANE & 15


def rows_of(w):
    """The two tile rows of wave w (lengths 16-w and w+1: 17 tiles for every wave)."""
    return (15 - w, w)


def slot_tile(w, s):
    """(I, J) of register slot s of wave w: tile row 15-w from its diagonal leftwards (slots 0..15-w), then tile row w
    ENDING on its diagonal (slot 16)."""
    if s <= 15 - w:
        return 15 - w, 15 - w - s
    return w, w + s - 16


def check_ownership():
    seen = {}
    for w in range(NW):
        for s in range(NS):
            I, Jt = slot_tile(w, s)
            assert 0 <= Jt <= I < NT, (w, s, I, Jt)
            assert (I, Jt) not in seen
            seen[(I, Jt)] = (w, s)
        assert [slot_tile(w, d) for d in (0, 16)] == [(r, r) for r in rows_of(w)]
    assert len(seen) == NT * (NT + 1) // 2
    return seen


def mfma(a, b, c):
    """v_mfma_f64_16x16x4_f64: a[lane] = A[m = lane & 15][k = lane >> 4], b[lane] = B[k = lane >> 4][n = lane & 15],
    c[lane][r] = C[4 r + (lane >> 4)][lane & 15]."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[J, G] = a
    B[G, J] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[4 * r + G, J]
    return out


def tiles_from_dense(M):
    acc = np.zeros((NW, NS, 64, 4))
    for w in range(NW):
        for s in range(NS):
            I, Jt = slot_tile(w, s)
            for r in range(4):
                acc[w, s, :, r] = M[16 * I + 4 * r + G, 16 * Jt + J]
    return acc


def dense_from_tiles(acc):
    M = np.zeros((256, 256))
    for w in range(NW):
        for s in range(NS):
            I, Jt = slot_tile(w, s)
            for r in range(4):
                M[16 * I + 4 * r + G, 16 * Jt + J] = acc[w, s, :, r]
    return np.tril(M) + np.tril(M, -1).T


def xl_off(c, g, kk):
    return c * CS + g * 4 + kk


def publish(acc, I0, trailing):
    """Phase 1: the panel X[k][c] = A[16 I0 + k][c] into LDS layout Xl[c][g][kk], k = 4 kk + g."""
    lds = np.full(256 * CS, np.nan)
    for w in range(NW):
        for s in range(NS):
            I, Jt = slot_tile(w, s)
            if I == I0:
                if trailing and Jt != I0:
                    continue
                for r in range(4):  # entry (k = 4 r + g, col 16 J + j): contiguous in kk = r
                    lds[xl_off(16 * Jt + J, G, r)] = acc[w, s, :, r]
            elif Jt == I0:
                for r in range(4):  # entry (i = 4 r + g, k = j) -> X[k = j][col = 16 I + 4 r + g]
                    lds[xl_off(16 * I + 4 * r + G, J & 3, J >> 2)] = acc[w, s, :, r]
    return lds


def load_b(lds, Jt, I0):
    """B operands of column tile Jt: bx[lane][kk] = X[4 kk + g][16 Jt + j] (minus the identity on the pivot columns)."""
    bx = np.zeros((64, 4))
    for kk in range(4):
        bx[:, kk] = lds[xl_off(16 * Jt + J, G, kk)]
        if Jt == I0:
            bx[:, kk] -= (4 * kk + G == J)
    return bx


def inv4_closed_form(c):
    """c[s][a] = P4[a][s] (columns of the symmetric 4 x 4 pivot block): closed form via 2 x 2 Schur complements.
    Returns the inverse and the positivity flag of the four sequential pivots (k_implicit_mfma_team.hip:226-252)."""
    a, b, e = c[0][0], c[0][1], c[1][1]
    b00, b01, b10, b11 = c[0][2], c[0][3], c[1][2], c[1][3]
    h, i2, jj = c[2][2], c[2][3], c[3][3]
    det_a = a * e - b * b
    ida = 1.0 / det_a
    ia00, ia01, ia11 = e * ida, -b * ida, a * ida
    t00, t01 = ia00 * b00 + ia01 * b10, ia00 * b01 + ia01 * b11
    t10, t11 = ia01 * b00 + ia11 * b10, ia01 * b01 + ia11 * b11
    s00 = h - (b00 * t00 + b10 * t10)
    s01 = i2 - (b00 * t01 + b10 * t11)
    s11 = jj - (b01 * t01 + b11 * t11)
    det_s = s00 * s11 - s01 * s01
    ids = 1.0 / det_s
    is00, is01, is11 = s11 * ids, -s01 * ids, s00 * ids
    ok = (a > 0) and (det_a > 0) and (s00 > 0) and (det_s > 0)
    u00, u01 = t00 * is00 + t01 * is01, t00 * is01 + t01 * is11
    u10, u11 = t10 * is00 + t11 * is01, t10 * is01 + t11 * is11
    p00 = ia00 + (u00 * t00 + u01 * t01)
    p01 = ia01 + (u00 * t10 + u01 * t11)
    p11 = ia11 + (u10 * t10 + u11 * t11)
    inv = np.array([[p00, p01, -u00, -u01], [p01, p11, -u10, -u11], [-u00, -u10, is00, is01], [-u01, -u11, is01, is11]])
    return inv, ok


def tile_sweep(t):
    """In-tile 4-wide symmetric sweep of the 16 x 16 pivot block held in accumulator layout t[lane][r]:
    returns T = -P^-1 in the same layout and the positivity flag."""
    t = t.copy()
    ok = True
    for R0 in range(4):
        scr = np.zeros(16 * 4)
        scr[J * 4 + G] = t[:, R0]  # panel of this sub-block: scr[c][s] = T[4 R0 + s][c]
        qv = np.stack([scr[J * 4 + s] for s in range(4)], 1)  # my column's four pivot-row entries
        cols = [[scr[(4 * R0 + b) * 4 + a] for a in range(4)] for b in range(4)]  # uniform: P4 columns
        inv, ok4 = inv4_closed_form(cols)
        ok = ok and ok4
        x4 = qv - (J[:, None] == 4 * R0 + np.arange(4)[None, :])  # X4[s][c] in x4[lane][s]
        nw = -(x4 @ inv.T)  # nw[lane][s] = -W4[s][c]
        a_op = nw[LANE, G]
        b_op = x4[LANE, G]
        t = mfma(a_op, b_op, t)
        t[:, R0] -= 2.0 * (J == 4 * R0 + G)
    return t, ok


def sweep(acc, trailing, nblk=NT):
    """Block-16 symmetric sweep.  Full: tiles end as -M^-1.  Trailing: block LDL^T (tile (K,K) = -P_K^-1, tile
    (I,K) = A_IK P_K^-1, I > K)."""
    acc = acc.copy()
    ok = True
    for I0 in range(nblk):
        lds = publish(acc, I0, trailing)
        t = np.stack([lds[xl_off(16 * I0 + J, G, r)] for r in range(4)], 1)  # every wave, redundantly
        t, okb = tile_sweep(t)
        ok = ok and okb
        for w in range(NW):
            nwrow = {}
            for I in rows_of(w):
                if trailing and I < I0:
                    continue
                c = np.zeros((64, 4))
                bx = load_b(lds, I, I0)
                for kk in range(4):  # -W[p][16 I + c] = sum_s T[p][s] X[s][16 I + c]
                    c = mfma(t[:, kk], bx[:, kk], c)
                nwrow[I] = c
            for s in range(NS):
                I, Jt = slot_tile(w, s)
                if trailing and Jt < I0:
                    continue
                bx = load_b(lds, Jt, I0)
                for kk in range(4):
                    acc[w, s] = mfma(nwrow[I][:, kk], bx[:, kk], acc[w, s])
                if I == I0 and Jt == I0:
                    for r in range(4):
                        acc[w, s, :, r] -= 2.0 * (J == 4 * r + G)
    return acc, ok


def matvec_full(acc, v):
    """y = S v for the symmetric matrix whose lower tiles are in acc (existing kernel's scheme: row sums of every
    tile row + column sums of the below-diagonal tiles)."""
    y = np.zeros(256)
    for w in range(NW):
        for s in range(NS):
            I, Jt = slot_tile(w, s)
            for r in range(4):
                np.add.at(y, 16 * I + 4 * r + G, acc[w, s, :, r] * v[16 * Jt + J])
                if I != Jt:
                    np.add.at(y, 16 * Jt + J, acc[w, s, :, r] * v[16 * I + 4 * r + G])
    return y


def ldlt_solve(acc, b):
    """Forward / diagonal / backward substitution with the trailing-sweep factors, in the kernel's order."""
    where = check_ownership()
    y = b.copy()
    z = np.zeros(256)
    for K in range(NT):
        # forward: rows I > K subtract T_IK y_K (row sums inside a tile: over the 16 lanes j)
        for I in range(K + 1, NT):
            w, s = where[(I, K)]
            for r in range(4):
                np.add.at(y, 16 * I + 4 * r + G, -acc[w, s, :, r] * y[16 * K + J])
        w, s = where[(K, K)]
        zk = np.zeros(16)
        for r in range(4):
            np.add.at(zk, 4 * r + G, -acc[w, s, :, r] * y[16 * K + J])
        z[16 * K:16 * K + 16] = zk
    u = z.copy()
    for K in range(NT - 1, -1, -1):
        for I in range(K + 1, NT):
            w, s = where[(I, K)]
            for r in range(4):  # column sums: over r in-lane and over g across rows
                np.add.at(u, 16 * K + J, -acc[w, s, :, r] * u[16 * I + 4 * r + G])
    return u


def main():
    rng = np.random.default_rng(0)
    check_ownership()
    worst = 0.0
    for dim in (256, 200, 77):
        A = rng.standard_normal((dim, dim))
        B = A @ A.T / dim + np.eye(dim)
        q = rng.standard_normal(dim)
        M = np.eye(256)
        M[:dim, :dim] = B + np.outer(q, q) / dim
        nblk = (dim + 15) // 16
        acc = tiles_from_dense(M)
        assert np.allclose(dense_from_tiles(acc), M)
        # in-tile sweep alone
        t0 = np.stack([M[4 * r + G, J] for r in range(4)], 1)
        t, ok = tile_sweep(t0)
        Pinv = np.zeros((16, 16))
        for r in range(4):
            Pinv[4 * r + G, J] = -t[:, r]
        e = np.abs(Pinv - np.linalg.inv(M[:16, :16])).max()
        assert ok and e < 1e-12, e
        # full inverse
        full, ok = sweep(acc, trailing=False, nblk=nblk)
        Minv = -dense_from_tiles(full)
        e1 = np.abs(Minv[:dim, :dim] - np.linalg.inv(M[:dim, :dim])).max()
        v = np.zeros(256)
        v[:dim] = rng.standard_normal(dim)
        e2 = np.abs(matvec_full(-full, v) - np.linalg.solve(M, v)).max()
        # trailing sweep + solves
        tr, ok2 = sweep(acc, trailing=True, nblk=nblk)
        u = ldlt_solve(tr, v)
        e3 = np.abs(u - np.linalg.solve(M, v)).max()
        print(f"dim {dim}: pivot-tile inverse {e:.1e}, full inverse {e1:.1e}, mat-vec {e2:.1e}, LDL^T solve {e3:.1e}")
        assert ok and ok2
        worst = max(worst, e1, e2, e3)
    # not positive definite -> flagged
    Mb = np.eye(256)
    Mb[5, 5] = -1.0
    _, ok = sweep(tiles_from_dense(Mb), trailing=True)
    assert not ok
    if worst > 1e-10:
        print("FAILED", worst)
        sys.exit(1)
    print("ok")


if __name__ == "__main__":
    main()
