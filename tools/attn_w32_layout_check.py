"""CPU check of the index algebra of csrc/attention_w32.hip (no GPU, no kernel code runs here).

The kernels rest on four hardware facts (MI355X guide + tests/test_gpu_ops.py::test_probe_*): the lane layouts of
v_mfma_f32_32x32x16_bf16 (A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31], D[(r&3)+8*(r>>2)+4*(l>>5)][l&31]), the gather of
ds_read_b64_tr_b16 (lane i of a 16-lane group addresses row R0+(i>>2), columns C0+4*(i&3).. and receives column C0+i of rows
R0..R0+3), the lane-linear LDS image of global_load_lds_dwordx4, and the LDS bank model (64 x 4-byte banks; ds_read_b128 in
16-lane groups {0-3,12-15,20-27},{4-11,16-19,28-31}(+32); the transposed read in two halves of 32 lanes).  Given those, this
script replays the kernels' address arithmetic in numpy and checks (1) every fragment read returns the elements the math needs,
(2) all fragment reads are bank-conflict free, (3) forward / dQ / dK,dV built from these pieces equal a plain softmax attention.
Run: python tools/attn_w32_layout_check.py
"""
import numpy as np

LOG2E = 1.4426950408889634
FMIN = np.float32(-3.4028234663852886e+38)


class WT:
    def __init__(self, HD, NW):
        self.HD, self.NW = HD, NW
        self.ROWB, self.CPR, self.TILE = HD * 2, HD // 8, 64 * HD * 2
        self.NPC = self.TILE // 1024 // NW

    def g(self, r):
        return ((((r >> 1) & 1) << 2) | ((r >> 2) & 3)) if self.HD == 64 else (((r & 3) << 2) | ((r >> 2) & 3))

    def off(self, r, c):
        return r * self.ROWB + ((c ^ self.g(r)) << 4)

    def dma_image(self, tile):
        """tile: [64, HD] array of element ids -> LDS image as array of elements indexed by byte/2."""
        lds = np.full(self.TILE // 2, -1, dtype=np.int64)
        for wid in range(self.NW):
            for j in range(self.NPC):
                for lane in range(64):
                    P = (wid * self.NPC + j) * 64 + lane
                    row, cph = P // self.CPR, P % self.CPR
                    col = (cph ^ self.g(row)) << 3
                    dst = ((wid * self.NPC + j) * 1024 + lane * 16) // 2
                    lds[dst:dst + 8] = tile[row, col:col + 8]
        assert (lds >= 0).all()
        return lds

    def fragA_addr(self, row, c):
        return self.off(row, c)

    def fragT_addrs(self, rbase, db, lane):
        i, gi = lane & 15, lane >> 4
        row = rbase + 4 * (gi >> 1) + (i >> 2)
        c = db * 4 + 2 * (gi & 1) + ((i >> 1) & 1)
        byte = (i & 1) * 8
        return self.off(row, c) + byte, self.off(row + 8, c) + byte


def tr_read(lds, addrs):
    """addrs[64] byte addresses -> out[64][4] per the probe-verified gather."""
    out = np.zeros((64, 4), dtype=lds.dtype)
    for grp in range(4):
        M = np.zeros((4, 16), dtype=lds.dtype)
        for rr in range(4):
            for k in range(4):
                a = addrs[grp * 16 + rr * 4 + k] // 2
                M[rr, 4 * k:4 * k + 4] = lds[a:a + 4]
        for i in range(16):
            out[grp * 16 + i] = M[:, i]
    return out


def banks_b128(addrs):
    """max conflict degree of a ds_read_b128 wave access under the documented lane groups."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups = groups + [[l + 32 for l in g_] for g_ in groups]
    worst = 1
    for g_ in groups:
        slots = {}
        for l in g_:
            s = (addrs[l] // 16) % 16
            slots.setdefault(s, set()).add(addrs[l])
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def banks_tr(addrs):
    worst = 1
    for half in range(2):
        b = {}
        for l in range(half * 32, half * 32 + 32):
            for w in range(2):
                bank = ((addrs[l] + 4 * w) // 4) % 64
                b.setdefault(bank, set()).add(addrs[l] + 4 * w)
        worst = max(worst, max(len(v) for v in b.values()))
    return worst


def check_layout(HD, NW):
    w = WT(HD, NW)
    tile = np.arange(64 * HD, dtype=np.int64).reshape(64, HD)
    lds = w.dma_image(tile)
    lanes = np.arange(64)
    for rb in range(2):
        for ds in range(HD // 16):
            addrs = [w.fragA_addr(rb * 32 + (l & 31), ds * 2 + (l >> 5)) for l in lanes]
            for l in lanes:
                got = lds[addrs[l] // 2: addrs[l] // 2 + 8]
                want = tile[rb * 32 + (l & 31), ds * 16 + 8 * (l >> 5): ds * 16 + 8 * (l >> 5) + 8]
                assert (got == want).all(), ("fragA", HD, rb, ds, l)
            assert banks_b128(addrs) == 1, ("fragA conflict", HD, rb, ds, banks_b128(addrs))
    for kk in range(2):
        for s in range(2):
            for db in range(HD // 32):
                a0 = [w.fragT_addrs(kk * 32 + 16 * s, db, l)[0] for l in lanes]
                a1 = [w.fragT_addrs(kk * 32 + 16 * s, db, l)[1] for l in lanes]
                lo, hi_ = tr_read(lds, a0), tr_read(lds, a1)
                for l in lanes:
                    d, hi = db * 32 + (l & 31), l >> 5
                    rows = [kk * 32 + 16 * s + 4 * hi + (j & 3) + 8 * (j >> 2) for j in range(8)]
                    want = tile[rows, d]
                    got = np.concatenate([lo[l], hi_[l]])
                    assert (got == want).all(), ("fragT", HD, kk, s, db, l, got, want)
                assert banks_tr(a0) == 1 and banks_tr(a1) == 1, ("fragT conflict", HD, banks_tr(a0), banks_tr(a1))
    print(f"layout HD={HD} NW={NW}: DMA image, row fragments, transposed fragments OK; bank-conflict free")


# ---------------------------------------------------------------- functional replay (one wave = 32 own rows), fp32 math
def mfma32(A, B, C):
    """A: [64 lanes][8] = A[m=l&31][k=8*(l>>5)+j]; B: [64][8] = B[k=8*(l>>5)+j][n=l&31]; C/D: [64][16]."""
    Am = np.zeros((32, 16), np.float64)
    Bm = np.zeros((16, 32), np.float64)
    for l in range(64):
        Am[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = A[l]
        Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = B[l]
    D = Am @ Bm
    out = C.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def rowfrag(T, rb, ds):      # fragA semantics on a [64, HD] tile
    return np.stack([T[rb * 32 + (l & 31), ds * 16 + 8 * (l >> 5): ds * 16 + 8 * (l >> 5) + 8] for l in range(64)])


def trfrag(T, rbase, db):    # fragT semantics
    out = np.zeros((64, 8))
    for l in range(64):
        hi = l >> 5
        rows = [rbase + 4 * hi + (j & 3) + 8 * (j >> 2) for j in range(8)]
        out[l] = T[rows, db * 32 + (l & 31)]
    return out


def ownfrag(X, row0, ds):    # register fragment of the wave's own rows
    return np.stack([X[row0 + (l & 31), ds * 16 + 8 * (l >> 5): ds * 16 + 8 * (l >> 5) + 8] for l in range(64)])


def acc_to_rows(acc_list):   # acc[db][l][r] -> X[own row][d]
    HD = 32 * len(acc_list)
    X = np.zeros((32, HD))
    for db, acc in enumerate(acc_list):
        for l in range(64):
            for r in range(16):
                X[l & 31, db * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)] = acc[l, r]
    return X


def reference(Q, K, V, kb, scale, fill):
    S = Q.shape[0]
    s = scale * (Q @ K.T) + kb[None, :]
    fut = np.arange(S)[None, :] > np.arange(S)[:, None]
    s = np.where(fut, np.where(kb[None, :] > FMIN, fill, kb[None, :]), s)
    s = np.where(kb[None, :] <= FMIN, FMIN, s)
    m = s.max(1, keepdims=True)
    e = np.exp(s - m)
    l = e.sum(1, keepdims=True)
    P = e / l
    return P @ V, m[:, 0], l[:, 0], P, (fut | (kb[None, :] <= FMIN))


def check_function(HD, S=128, pad=(), seed=0):
    rng = np.random.default_rng(seed)
    Q, K, V, dO = (rng.standard_normal((S, HD)) for _ in range(4))
    scale = 1.0 / np.sqrt(HD)
    pos = np.arange(S, dtype=np.float64)
    kb = 0.0625 * pos
    for k in pad:
        kb[k] = FMIN
    first_valid = 0
    while first_valid < S and kb[first_valid] <= FMIN:
        first_valid += 1
    O_ref, m_ref, l_ref, P, masked = reference(Q, K, V, kb, scale, float(FMIN))
    dP = dO @ V.T
    delta = (dO * O_ref).sum(1)
    dS = np.where(masked, 0.0, P * (dP - delta[:, None]))
    dQ_ref, dK_ref, dV_ref = scale * dS @ K, scale * dS.T @ Q, P.T @ dO
    c, kb2 = scale * LOG2E, np.where(kb <= FMIN, float(FMIN), kb * LOG2E)
    NDS, NDB = HD // 16, HD // 32
    hi = np.arange(64) >> 5
    l32 = np.arange(64) & 31

    def key_of(kv0, kk, r):
        return kv0 + kk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi

    # ---- forward + dQ for every 32-row query block
    O = np.zeros((S, HD)); M = np.zeros(S); L = np.zeros(S); dQ = np.zeros((S, HD))
    for q0w in range(0, S, 32):
        allk = first_valid > q0w - (q0w % 256)
        ntiles = S // 64 if allk else min(S, (q0w // 256) * 256 + 256) // 64
        my_last = ntiles - 1 if allk else min(ntiles - 1, (q0w + 31) // 64)
        qf = [ownfrag(Q, q0w, ds) for ds in range(NDS)]
        o = [np.zeros((64, 16)) for _ in range(NDB)]
        m = np.full(64, -np.inf); lsum = np.zeros(64)
        for t in range(my_last + 1):
            kv0 = t * 64
            Kt, Vt = K[kv0:kv0 + 64], V[kv0:kv0 + 64]
            x = [np.zeros((64, 16)), np.zeros((64, 16))]
            for ds in range(NDS):
                for kk in range(2):
                    x[kk] = mfma32(rowfrag(Kt, kk, ds), qf[ds], x[kk])
            for kk in range(2):
                for r in range(16):
                    kbv = kb2[key_of(kv0, kk, r)]
                    v = x[kk][:, r] * c + kbv
                    v = np.where(kbv <= FMIN, float(FMIN), v)      # fp32 absorption of the fma
                    if kv0 + 63 > q0w:
                        thr = q0w + l32 - kv0 - 4 * hi
                        cc = kk * 32 + 8 * (r >> 2) + (r & 3)
                        v = np.where(cc > thr, np.where(kbv > FMIN, float(FMIN), kbv), v)
                    x[kk][:, r] = v
            mx = np.maximum(x[0].max(1), x[1].max(1))
            mx = np.maximum(mx, mx[np.arange(64) ^ 32])
            m_new = np.maximum(m, mx)
            alpha = np.exp2(m - m_new)
            for kk in range(2):
                x[kk] = np.exp2(x[kk] - m_new[:, None])
            lsum = lsum * alpha + x[0].sum(1) + x[1].sum(1)
            for db in range(NDB):
                o[db] *= alpha[:, None]
            m = m_new
            for kk in range(2):
                for s in range(2):
                    pb = x[kk][:, 8 * s:8 * s + 8]
                    for db in range(NDB):
                        o[db] = mfma32(trfrag(Vt, kk * 32 + 16 * s, db), pb, o[db])
        lsum = lsum + lsum[np.arange(64) ^ 32]
        Ow = acc_to_rows([o_ / lsum[:, None] for o_ in o])
        O[q0w:q0w + 32] = Ow
        M[q0w:q0w + 32] = np.where(m[:32] <= FMIN, float(FMIN), m[:32] / LOG2E)
        L[q0w:q0w + 32] = lsum[:32]
        # dQ
        general = first_valid > 0
        nt_dq = min(S, (q0w // 256) * 256 + 256) // 64
        last_dq = min(nt_dq - 1, (q0w + 31) // 64)
        gf = [ownfrag(dO, q0w, ds) for ds in range(NDS)]
        mm = M[q0w + l32]; m2 = np.where(mm <= FMIN, float(FMIN), mm * LOG2E); il = 1.0 / L[q0w + l32]
        dl = (dO[q0w + l32] * O[q0w + l32]).sum(1)
        dq = [np.zeros((64, 16)) for _ in range(NDB)]
        for t in range(last_dq + 1):
            kv0 = t * 64
            Kt, Vt = K[kv0:kv0 + 64], V[kv0:kv0 + 64]
            x = [np.zeros((64, 16)), np.zeros((64, 16))]; y = [np.zeros((64, 16)), np.zeros((64, 16))]
            for ds in range(NDS):
                for kk in range(2):
                    x[kk] = mfma32(rowfrag(Kt, kk, ds), qf[ds], x[kk])
                    y[kk] = mfma32(rowfrag(Vt, kk, ds), gf[ds], y[kk])
            maskt = general or (kv0 + 63 > q0w)
            thr = q0w + l32 - kv0 - 4 * hi
            for kk in range(2):
                for r in range(16):
                    kbv = kb2[key_of(kv0, kk, r)]
                    s2 = np.where(kbv <= FMIN, float(FMIN), x[kk][:, r] * c + kbv)
                    with np.errstate(over="ignore", invalid="ignore"):
                        pr = np.exp2(s2 - m2) * il
                        d = pr * (y[kk][:, r] - dl)
                    if maskt:
                        cc = kk * 32 + 8 * (r >> 2) + (r & 3)
                        d = np.where((kbv > FMIN) & (cc <= thr), d, 0.0)
                    y[kk][:, r] = d
            for kk in range(2):
                for s in range(2):
                    for db in range(NDB):
                        dq[db] = mfma32(trfrag(Kt, kk * 32 + 16 * s, db), y[kk][:, 8 * s:8 * s + 8], dq[db])
        dQ[q0w:q0w + 32] = acc_to_rows(dq) * scale
    assert np.allclose(O, O_ref, atol=1e-9), np.abs(O - O_ref).max()
    assert np.allclose(L, l_ref, rtol=1e-9) and np.allclose(M, m_ref, rtol=1e-9)
    assert np.allclose(dQ, dQ_ref, atol=1e-9), np.abs(dQ - dQ_ref).max()

    # ---- dK, dV for every 32-row key block
    dK = np.zeros((S, HD)); dV = np.zeros((S, HD))
    allq = first_valid > 0
    m2S = np.where(M <= FMIN, float(FMIN), M * LOG2E); ilS = 1.0 / L
    for k0w in range(0, S, 32):
        my_first = 0 if allq else k0w // 64
        kf = [ownfrag(K, k0w, ds) for ds in range(NDS)]
        vf = [ownfrag(V, k0w, ds) for ds in range(NDS)]
        key = k0w + l32
        key_pad = kb[key] <= FMIN
        kb_lane = kb2[key]
        dk = [np.zeros((64, 16)) for _ in range(NDB)]; dv = [np.zeros((64, 16)) for _ in range(NDB)]
        for t in range(my_first, S // 64):
            Qt, Gt = Q[t * 64:t * 64 + 64], dO[t * 64:t * 64 + 64]
            x = [np.zeros((64, 16)), np.zeros((64, 16))]; y = [np.zeros((64, 16)), np.zeros((64, 16))]
            for ds in range(NDS):
                for qq in range(2):
                    x[qq] = mfma32(rowfrag(Qt, qq, ds), kf[ds], x[qq])
                    y[qq] = mfma32(rowfrag(Gt, qq, ds), vf[ds], y[qq])
            maskt = allq or (t * 64 < k0w + 31)
            thr = np.where(key_pad, 0x7fffffff, k0w + l32 - t * 64 - 4 * hi)
            for qq in range(2):
                for r in range(16):
                    qi = t * 64 + qq * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi
                    s2 = np.where(key_pad, float(FMIN), x[qq][:, r] * c + kb_lane)
                    msk = np.zeros(64, bool)
                    if maskt:
                        cc = qq * 32 + 8 * (r >> 2) + (r & 3)
                        msk = cc < thr
                        s2 = np.where(msk, float(FMIN), s2)
                    with np.errstate(over="ignore", invalid="ignore"):
                        pr = np.exp2(s2 - m2S[qi]) * ilS[qi]
                        d = pr * (y[qq][:, r] - delta[qi])
                    if maskt:
                        d = np.where(msk, 0.0, d)
                    x[qq][:, r] = pr; y[qq][:, r] = d
            for qq in range(2):
                for s in range(2):
                    for db in range(NDB):
                        dv[db] = mfma32(trfrag(Gt, qq * 32 + 16 * s, db), x[qq][:, 8 * s:8 * s + 8], dv[db])
                        dk[db] = mfma32(trfrag(Qt, qq * 32 + 16 * s, db), y[qq][:, 8 * s:8 * s + 8], dk[db])
        dKw = acc_to_rows(dk) * scale
        dKw[key_pad[:32]] = 0.0
        dK[k0w:k0w + 32] = dKw
        dV[k0w:k0w + 32] = acc_to_rows(dv)
    assert np.allclose(dK, dK_ref, atol=1e-9), np.abs(dK - dK_ref).max()
    assert np.allclose(dV, dV_ref, atol=1e-9), np.abs(dV - dV_ref).max()
    print(f"function HD={HD} S={S} pad={list(pad)[:4]}{'...' if len(pad) > 4 else ''}: forward, dQ, dK, dV == reference")


if __name__ == "__main__":
    for HD, NW in ((64, 8), (128, 8), (128, 4), (64, 4)):
        check_layout(HD, NW)
    check_function(64, 128)
    check_function(64, 320, pad=range(0, 70))          # left padding: all-masked rows, uniform over all keys
    check_function(64, 192, pad=range(150, 192))       # right padding
    check_function(128, 128, pad=(3, 77))
