"""Index-level emulation of csrc/attention_d512.hip on the CPU (no GPU, no kernel code executed): the LDS fill with its source-side
XOR swizzles, the fragment addresses, the permuted K rows, the MFMA 32x32x16 operand / result lane layouts, the tail mask, the two
passes and the output store are restated with the kernel's own formulas and must reproduce softmax(Q K^T / sqrt(512)) V in float64.

This pins the kernel's ADDRESS ALGEBRA (what a blind bug would most likely be); what it cannot cover -- LDS-DMA semantics, barriers,
counted vmcnt, the compiler -- is covered by tests/test_attn_d512_gpu.py on hardware.  The MFMA lane layout assumed here
(A: row = lane & 31, k = 8 * (lane >> 5) + i; B: column = lane & 31, same k; D: column = lane & 31, row = (r & 3) + 8 * (r >> 2) +
4 * (lane >> 5)) is the one csrc/attention.hip relies on and the GPU suite validates."""
import numpy as np
import pytest

NW, KT = 4, 32


def _mfma(a_frag, b_frag, c):
    a, b = np.zeros((32, 16)), np.zeros((16, 32))
    for lane in range(64):
        for i in range(8):
            a[lane & 31, 8 * (lane >> 5) + i] = a_frag[lane][i]
            b[8 * (lane >> 5) + i, lane & 31] = b_frag[lane][i]
    d = a @ b
    out = c.copy()
    for lane in range(64):
        for r in range(16):
            out[lane][r] += d[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
    return out


def _emulate(q_mat, k_mat, v_mat, ldvt):
    tq, tk = q_mat.shape[0], k_mat.shape[0]
    vt = np.zeros((512, ldvt))
    vt[:, :tk] = v_mat.T
    c = 512 ** -0.5 * 1.4426950408889634
    out = np.zeros((tq, 512))
    nt = (tk + KT - 1) // KT
    qb_rows = 32 * NW
    lanes = range(64)

    def fill(t):   # stage_k / stage_v: LDS position <- global source chunk
        s_k, s_v = np.zeros((32, 64, 8)), np.zeros((512, 4, 8))
        for wave in range(NW):
            for i in range(32 // NW):
                row = i * NW + wave
                kr = min(t * KT + row, tk - 1)
                for lane in lanes:
                    src = ((lane << 4) ^ ((row & 15) << 4)) >> 4          # lane16 ^ ((row & 15) << 4)
                    s_k[row, lane] = k_mat[kr, src * 8:src * 8 + 8]
                ins = i * NW + wave
                for lane in lanes:
                    voff_row, ch = lane >> 2, (lane & 3) ^ ((lane >> 4) & 3)   # voff = (lane >> 2) * ldvt * 2 + (ch << 4)
                    row_v = ins * 16 + voff_row
                    assert ins * 1024 + lane * 16 == row_v * 64 + (lane & 3) * 16   # LDS slot of this lane
                    s_v[row_v, lane & 3] = vt[row_v, t * KT + ch * 8:t * KT + ch * 8 + 8]
        return s_k, s_v

    for qb in range((tq + qb_rows - 1) // qb_rows):
        for wave in range(NW):
            q_of = [qb * qb_rows + wave * 32 + (lane & 31) for lane in lanes]
            qc = [min(q, tq - 1) for q in q_of]
            qf = [[q_mat[qc[lane], 16 * ks + 8 * (lane >> 5):16 * ks + 8 * (lane >> 5) + 8] for lane in lanes] for ks in range(32)]

            def scores(s_k, c0, t):
                s0, s1 = np.zeros((64, 16)), np.zeros((64, 16))
                for ks in range(32):
                    kfr = []
                    for lane in lanes:
                        l31, half = lane & 31, lane >> 5
                        krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1)
                        kx = (krow & 15) ^ half
                        kfr.append(s_k[krow, ((2 * ks) & 48) | (((2 * ks) & 15) ^ kx)])
                    if ks & 1:
                        s1 = _mfma(kfr, qf[ks], s1)
                    else:
                        s0 = _mfma(kfr, qf[ks], s0)
                s = (s0 + s1) + c0[:, None]
                if t == nt - 1:
                    for lane in lanes:
                        for r in range(16):
                            if t * KT + 16 * (r >> 3) + 8 * (lane >> 5) + (r & 7) > tk - 1:
                                s[lane][r] = -np.inf
                return s

            mx = np.full(64, -np.inf)
            for t in range(nt):                                     # pass 1
                mx = np.maximum(mx, scores(fill(t)[0], np.zeros(64), t).max(axis=1))
            nm = -np.array([max(mx[lane], mx[lane ^ 32]) for lane in lanes])
            o = [np.zeros((64, 16)) for _ in range(16)]
            l_run = np.zeros(64)
            for t in range(nt):                                     # pass 2
                s_k, s_v = fill(t)
                p = np.exp2(scores(s_k, nm, t) * c)
                l_run += p.sum(axis=1)
                pf = [[p[lane][8 * j:8 * j + 8] for lane in lanes] for j in range(2)]
                for db in range(16):
                    for j in range(2):
                        vfr = [s_v[db * 32 + (lane & 31), (2 * j) ^ (((lane & 31) >> 2) & 3) ^ (lane >> 5)] for lane in lanes]
                        o[db] = _mfma(vfr, pf[j], o[db])
            l_tot = [l_run[lane] + l_run[lane ^ 32] for lane in lanes]
            for lane in lanes:
                if q_of[lane] < tq:
                    for db in range(16):
                        for rg in range(4):
                            for e in range(4):
                                out[q_of[lane], db * 32 + 8 * rg + 4 * (lane >> 5) + e] = o[db][lane][rg * 4 + e] / l_tot[lane]
    return out


@pytest.mark.parametrize("tq,tk,ldvt", [(40, 70, 96), (130, 33, 64), (5, 1, 32)])
def test_attn_d512_address_algebra(tq, tk, ldvt):
    rng = np.random.default_rng(tq * 1000 + tk)
    q, k, v = rng.standard_normal((tq, 512)), rng.standard_normal((tk, 512)), rng.standard_normal((tk, 512))
    s = (q @ k.T) * 512 ** -0.5
    p = np.exp(s - s.max(axis=1, keepdims=True))
    ref = (p / p.sum(axis=1, keepdims=True)) @ v
    assert np.abs(_emulate(q, k, v, ldvt) - ref).max() < 1e-12
