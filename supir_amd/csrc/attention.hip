// Flash attention forward, head dim 64, bf16 in / fp32 online softmax / bf16 out, for gfx950 (MI355X).
//
// Replaces xformers.ops.memory_efficient_attention / F.scaled_dot_product_attention at
// sgm/modules/attention.py:273-277,357-359 (CrossAttention / MemoryEfficientCrossAttention) and the
// ZeroCrossAttn call at SUPIR/modules/SUPIR_v0.py:146.  softmax(Q K^T / sqrt(64)) V, no mask, no dropout.
//
// Layout (chosen so no transpose kernel is ever needed):
//   Q  [B][Tq][ldq]  head h at columns h*64..h*64+63   (exactly what the to_q GEMM writes: 'b n (h d)')
//   K  [B][Tk][ldk]  same
//   Vt [B][H*64][ldvt]  V TRANSPOSED per batch (row = h*64+d, column = key), written by the to_v GEMM's
//                       transposed epilogue; ldvt >= round_up(Tk,64) and the padding is finite (zero)
//   O  [B][Tq][ldo]
// One workgroup = 4 waves = 128 query rows of one (batch, head); K / Vt tiles of 64 keys are staged through LDS by
// global_load_lds (3-deep ring, counted vmcnt) and shared by the 4 waves.
// MFMA 32x32x16 with swapped operands: S^T = K.Q^T puts one query row per lane (the row maximum needs one
// v_permlane32_swap), and P feeds the PV MFMA straight from those registers: the key order inside each 16-wide k block is
// arranged (K rows read with bits 2/3 of the row index swapped) so that it is one 16-byte chunk of a Vt row: no
// cross-lane exchange, conflict-free ds_read_b128 on both operands.
#include "kernels.h"
#include <stdlib.h>


#ifdef SUPIR_ATTN_TIMELINE
// tools/probes/attn_timeline.hip only: per-wave s_memtime breakdown of the KV loop (never defined in the product build)
__device__ unsigned long long* g_atl_buf;
extern "C" void supir_atl_set(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_atl_buf), &p, sizeof(p)); }
#define ATL(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define ATL(var)
#endif

template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------------------------------------
// Software pipeline.  Run strictly one after the other, Q.K^T, softmax and P.V of a 64-key tile cost 2470 cycles per wave
// for 512 cycles of MFMA (profiles/r01/attn_timeline_after.log, profiles/r02/attn_timeline_unpipelined.log): a wave issues
// in order, one VALU instruction per >= 4 cycles (8 for v_exp_f32) and 12.5 per MFMA, and up to five plain VALU
// instructions are free beside each 32-cycle MFMA (tools/probes/issue_probe.hip, profiles/r02/issue_probe.log).  So the
// unit here is a HALF tile g (32 keys) and every MFMA is issued with independent VALU work of the neighbouring half tile:
//     slots 1-4:  S(g+1) = K(g+1).Q^T - m   (4 chained MFMAs, C operand = -m)  |  P(g) = exp2(S(g)), row sums, bf16 pack
//     slots 5-8:  O += V^T(g).P(g)          (4 MFMAs)                          |  row maximum of S(g+1)
// K / V^T fragments are ds_read one stage ahead of their MFMAs.  Because S(t+1, half 0) is produced during tile t, tile t+1
// must already be in LDS while tile t is being consumed: an S-deep ring (S >= 3), one barrier per tile at which tile t+1
// is awaited and the loads of tile t+S-1 are issued.  Measured (profiles/r02/attn_pipelined_vs_unpipelined_timing.log):
// 28.9 -> 24.3 us at (B2, H20, 1024^2), 138 -> 114 us at (B2, H10, 4096^2) = 755 TFLOP/s.
template <bool V>
struct attn_flag {
    static constexpr bool value = V;
};

__device__ __forceinline__ float xhalf_max(float x) {
    // max over lanes l and l^32: one v_permlane32_swap, no LDS round trip in the dependent chain
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// NP = 2 (supir_flash_attn_d64_grouped): two independent attention problems in one grid, problem q on XCDs [4 q, 4 q + 4): at 1024
// tokens a problem is 320 workgroups on 256 CUs -- 1.25 rounds; two of them back to back fill 2.5 rounds instead of 2 x 2.
// R4 (round 4, the default): no tile loads past the last tile, and the output staged through LDS for row-contiguous 16-byte stores;
// R4 = false is the round-3 kernel, kept selectable for A/B runs (tools-only knob 3).
// DS (round 4, second half): the row sum l as v_dot2c of the PACKED bf16 probabilities -- one instruction per two terms instead of an fp32
// add per term (the exponential slots are VALU-issue bound: 4 exp + 4 add + 2 cvt_pk per MFMA) -- which also makes l the sum of the
// probabilities P.V really multiplies with.  DS = false: the fp32 sum of the unrounded exponentials (rounds 1-3; tools-only knob 3 = 4,
// round-4 forms only).  Measured (profiles/r04/micro_flash_attention_row_sum_dot2_vs_fp32.log): 21.9 -> 20.8 us at (B2, H20, 1024^2),
// 104.6 -> 103.4 at (B2, H10, 4096^2), error vs fp32 SDPA equal or slightly lower (2.866e-3 vs 2.872e-3; 3.87e-3 vs 4.00e-3 at 3 x the logits).
template <int S, int NW, bool PRE, int NP = 1, bool R4 = true, bool DS = true>
__global__ __launch_bounds__(64 * NW, 3) void attn_d64_pipe_kernel(const AttnArgsN<NP> pp) {
    static_assert(S >= 3, "tile t+1 is read while tile t is live and tile t+2 is in flight");
    __shared__ __attribute__((aligned(16))) char smem[S * 16384];
    constexpr int NX = 8 / NP;
    const int prob = NP == 1 ? 0 : (int)(blockIdx.x & 7) / NX;   // wave-uniform
    const AttnArgs& p = pp.p[prob];
    constexpr int NT = 64 * NW, QB = 32 * NW, LPT = 512 / NT, NL = 2 * LPT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    const int nqb = (p.Tq + QB - 1) / QB;
    int id;
    if constexpr (NP == 1) {
        id = xcd_remap(blockIdx.x, nqb * p.H * p.B);
    } else {   // the problem's workgroups in contiguous runs over its NX XCDs; the grid is rounded up to whole rows of 8 blocks
        const int nwg = nqb * p.H * p.B, qn = nwg / NX, rn = nwg - qn * NX;
        const int vx = (int)blockIdx.x & (NX - 1), vidx = (int)blockIdx.x >> 3;
        if (vidx >= qn + (vx < rn ? 1 : 0)) return;
        id = (vx < rn ? vx * (qn + 1) : rn * (qn + 1) + (vx - rn) * qn) + vidx;
    }
    const int bh = id / nqb, qb = id - bh * nqb;
    const int b = bh / p.H, h = bh - b * p.H;
    const char* Kb = (const char*)(p.K + (size_t)b * p.Tk * p.ldk + h * 64);
    const char* Vb = (const char*)(p.Vt + ((size_t)b * p.H + h) * 64 * p.ldvt);

    const int q = qb * QB + wave * 32 + l31;
    const bool q_ok = q < p.Tq;
    const int qc = q_ok ? q : p.Tq - 1;
    const bf16_t* Qp = p.Q + ((size_t)b * p.Tq + qc) * p.ldq + h * 64 + 8 * half;
    const float c = p.scale_log2e;
    const float pmul = PRE ? 1.0f : c;   // PRE = false: S stays in raw q.k units and the exponent pays one multiply per element
    bf16x8 qf[4];   // loaded after the first K / V^T tiles have been requested (their latencies then overlap)

    // ---- loader.  Source address = wave-uniform tile base (SGPR pair) + a per-lane 32-bit byte offset that does not
    // change from tile to tile, so a tile costs no address VALU; only the last (ragged) tile of K needs its rows clamped,
    // and that is a second, precomputed set of offsets chosen by a wave-uniform select.
    const int nt = (p.Tk + 63) >> 6;
    const int lrow = tid >> 3;
    const int lchunk = (tid & 7) ^ ((tid >> 4) & 7);
    unsigned kofs[LPT], kofs_last[LPT], vofs[LPT];
    const int last_rows = p.Tk - 1 - (nt - 1) * 64;   // highest valid row of the last tile
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int r = i * (NT / 8) + lrow;
        kofs[i] = (unsigned)(r * p.ldk * 2 + lchunk * 16);
        kofs_last[i] = (unsigned)((r < last_rows ? r : last_rows) * p.ldk * 2 + lchunk * 16);
        vofs[i] = (unsigned)(r * p.ldvt * 2 + lchunk * 16);
    }
    const size_t k_tile_bytes = (size_t)64 * p.ldk * 2;
    auto stage_one = [&](int t, int soff, int i) {   // soff = ring slot of tile t, in bytes
        // past the end: nothing to load.  (Until round 4 the last tile was re-loaded into a ring slot nobody reads, to keep the number
        // of loads in flight per iteration uniform -- but with S = 3 every tile start waits for vmcnt(0) anyway, so the reloads only
        // cost 2 x 16 KB of L2 -> LDS traffic per workgroup and a drain at the end: 25 % of the loads of a 77-key launch.)
        if (R4 && S == 3 && t >= nt) return;
        const int tc = t < nt - 1 ? t : nt - 1;   // (S > 3, counted waits: reload the last tile into a ring slot nobody reads any more)
        char* sK = smem + soff;
        char* sV = sK + 8192;
        if (i < LPT) {
            const unsigned off = t < nt - 1 ? kofs[i] : kofs_last[i];
            glds16(Kb + tc * k_tile_bytes + off, sK + (i * NT + wave * 64) * 16);
        } else {
            glds16(Vb + (size_t)tc * 128 + vofs[i - LPT], sV + ((i - LPT) * NT + wave * 64) * 16);
        }
    };

    // ---- LDS fragment addresses (per lane, tile independent); the ring slot is a scalar added per tile.
    // K fragment rows are read PERMUTED: MFMA row i of S^T holds key swap_bits(2,3)(i).  With that, the 8 P values a lane
    // owns per 16-key block (rows 4*half + (r&3) + 8*(r>>2)) are the 8 CONSECUTIVE keys 16*kb + 8*half + 0..7, i.e. exactly
    // one 16-byte chunk of a V^T row: the PV operand is a conflict-free ds_read_b128 and no lane exchange is needed.
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int ksw = (krow >> 1) & 7, vsw = (l31 >> 1) & 7;
    int kaddr[4], vaddr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        kaddr[i] = krow * 128 + (((2 * i + half) ^ ksw) * 16);          // + hf*4096: K half tile hf, k step i
        vaddr[i] = 8192 + l31 * 128 + (((2 * i + half) ^ vsw) * 16);    // + df*4096: keys 16*i + 8*half .. +7 of V^T rows df*32 + l31
    }
    auto load_k = [&](bf16x8 (&kf)[4], int boff, int hf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8*)(smem + boff + hf * 4096 + kaddr[ks]);
    };
    auto load_v = [&](bf16x8 (&vf)[4], int boff, int hf) {   // vf[kb*2+df]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int df = 0; df < 2; ++df) vf[kb * 2 + df] = *(const bf16x8*)(smem + boff + df * 4096 + vaddr[2 * hf + kb]);
    };

    f32x16 o[2], nm;   // nm = -m_run in every element: the C operand of the first Q.K^T MFMA of each half tile
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = nm[r] = 0.f;
    float l_run = 0.f;
    const bool tail_mask = (p.Tk & 63) != 0;
    const bool last_half_empty = tail_mask && (p.Tk & 63) <= 32 && !p.causal;
    const int klim = p.causal ? (q < p.Tk - 1 ? q : p.Tk - 1) : p.Tk - 1;   // last visible key of this lane's query
    // four S values (already relative to the running maximum) of a half tile that starts at key k0: mask, fold into mx
    auto max4 = [&](f32x16& s, float mx, int k0, int j, bool masked) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * j + e;
            if (masked && k0 + 16 * (r >> 3) + 8 * half + (r & 7) > klim) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        return mx;
    };
    // s holds S' - m_ref and d = its row maximum (over both lane halves).  m_ref is a REFERENCE, not the exact running maximum:
    // it is moved (s, nm, O and l rebased) only when some row exceeds it by more than 2^8, so P stays <= 256 -- exact in
    // fp32, and bf16 keeps its relative precision at any magnitude.  An exact running maximum would take this branch whenever
    // any of the wave's 32 rows sees a new maximum: most half tiles of the first few hundred keys (86 % at key 512 for i.i.d.
    // logits), ~70 VALU instructions each time (measured: 29.4 -> 22.9 us at 1024^2, tools/probes/attn_timeline.hip).
    constexpr float REBASE_AT = 8.0f;
    auto rebase = [&](f32x16& s, float d) {
        if (__builtin_amdgcn_ballot_w64(d * pmul > REBASE_AT) != 0) {
            asm volatile("" ::: "memory");
            d = fmaxf(d, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-d * pmul);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] -= d;
                nm[r] -= d;
                o[0][r] *= alpha;
                o[1][r] *= alpha;
            }
        }
    };
    // P values 4*j .. 4*j+3 of a half tile: exp2, row sum, bf16 pack into the P.V operand.  The empty asm pins the
    // exponentials to the slot they are written in (otherwise they are sunk to their first use, behind the MFMAs they are
    // meant to run under).  Plain (unpacked) fp32 VALU on purpose: v_pk_* beside MFMAs issues slower than two scalar ops
    // (tools/probes/issue_probe.hip).
    auto exp4 = [&](const f32x16& s, bf16x8 (&pf)[2], int j) {
        if constexpr (DS) {
            float pv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pv[e] = PRE ? __builtin_amdgcn_exp2f(s[4 * j + e]) : __builtin_amdgcn_exp2f(s[4 * j + e] * c);
                asm volatile("" : "+v"(pv[e]));
            }
            u32x4 w = __builtin_bit_cast(u32x4, pf[j >> 1]);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const uint32_t pk = f2bf_pk(pv[2 * h2], pv[2 * h2 + 1]);
                l_run = pair_sum_acc(pk, l_run);
                w[2 * (j & 1) + h2] = pk;
            }
            pf[j >> 1] = __builtin_bit_cast(bf16x8, w);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * j + e;
                float pv = PRE ? __builtin_amdgcn_exp2f(s[r]) : __builtin_amdgcn_exp2f(s[r] * c);
                asm volatile("" : "+v"(pv));
                l_run += pv;
                pf[r >> 3][r & 7] = (bf16_t)pv;
            }
        }
    };

#ifdef SUPIR_ATTN_TIMELINE
    unsigned long long tl_sync = 0, tl_a0 = 0, tl_b0 = 0, tl_a1 = 0, tl_b1 = 0;
#endif
    ATL(tl_t0);
    // ---- prologue: tiles 0 .. S-2 in flight, tiles 0 and 1 awaited, S(0, half 0) and its maximum computed un-overlapped
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
#pragma unroll
        for (int i = 0; i < NL; ++i) stage_one(t, t * 16384, i);
    // Q carries the softmax scale (log2 units) from here on: S' = (c Q).K^T, so that P = exp2(S' - m) needs no multiply per
    // element; the MFMA's C operand supplies the "- m" (nm).  One extra bf16 rounding of Q (2^-9 relative, averaged over the
    // 64-term dot product) against 32 VALU instructions per lane per tile.
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 raw = *(const bf16x8*)(Qp + 16 * ks);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ks][e] = PRE ? (bf16_t)((float)raw[e] * c) : raw[e];
    }
    bf16x8 kA[4], vA[4], pf[2];
    f32x16 sa, sb;
    attn_wait_vmcnt<(S - 3) * NL>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_k(kA, 0, 0);
    sa = nm;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = SUPIR_MFMA_32x32x16(kA[ks], qf[ks], sa, 0, 0, 0);
    load_k(kA, 0, 1);
    {
        const bool mk = (nt == 1 && tail_mask) || p.causal;
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = max4(sa, mx, 0, j, mk);
        mx = xhalf_max(mx);   // finite: key 0 is visible to every query
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sa[r] -= mx;
            nm[r] = -mx;
        }
    }

    // One tile.  MASKED (compile time) = this iteration produces half tiles that may hold keys past the end / past the
    // diagonal: only the last two iterations of a ragged launch, or every iteration of a causal one, so that the common
    // loop body is straight-line code between the two (rare) rebase branches.
    int boff = 0;   // ring slot of the current tile (bytes), carried as a scalar
    auto tile = [&](const int t, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
        const int nboff = __builtin_amdgcn_readfirstlane(boff + 16384 == S * 16384 ? 0 : boff + 16384);
        const int soff = __builtin_amdgcn_readfirstlane(boff == 0 ? (S - 1) * 16384 : boff - 16384);   // slot of tile t-1 = of tile t+S-1
        ATL(tl_0);
        if (t > 0) {
            // tile t+1 landed (issued one tile ago); every wave is done with tile t-1, whose ring slot takes tile t+S-1
            attn_wait_vmcnt<(S - 3) * NL>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        ATL(tl_1);
        // ================= half 0: P(t,0) from sa, S(t,1) into sb =================
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sb = SUPIR_MFMA_32x32x16(kA[j], qf[j], j == 0 ? nm : sb, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 0) load_v(vA, boff, 0);   // after the MFMA: it must not wait for these reads, only for kA's
            exp4(sa, pf, j);
#pragma unroll
            for (int i = j * NL / 4; i < (j + 1) * NL / 4; ++i) stage_one(t + S - 1, soff, i);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef SUPIR_ATTN_TIMELINE
        asm volatile("s_nop 0" ::"v"(sb[0]), "v"(pf[1]));
#endif
        ATL(tl_2);
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j & 1] = SUPIR_MFMA_32x32x16(vA[j], pf[j >> 1], o[j & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 0) load_k(kA, nboff, 0);   // past the last tile: a ring slot nobody waits for; its products are never used
            mx = max4(sb, mx, t * 64 + 32, j, MASKED);
            __builtin_amdgcn_sched_barrier(0);
        }
        rebase(sb, xhalf_max(mx));
#ifdef SUPIR_ATTN_TIMELINE
        asm volatile("s_nop 0" ::"v"(o[0][0]), "v"(o[1][15]), "v"(sb[0]));
#endif
        ATL(tl_3);
        // ================= half 1: P(t,1) from sb, S(t+1,0) into sa =================
        if (MASKED && t == nt - 1 && last_half_empty) return;   // nothing but masked keys left (Tk % 64 in 1..32): P = 0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sa = SUPIR_MFMA_32x32x16(kA[j], qf[j], j == 0 ? nm : sa, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 0) load_v(vA, boff, 1);
            exp4(sb, pf, j);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef SUPIR_ATTN_TIMELINE
        asm volatile("s_nop 0" ::"v"(sa[0]), "v"(pf[1]));
#endif
        ATL(tl_4);
        mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j & 1] = SUPIR_MFMA_32x32x16(vA[j], pf[j >> 1], o[j & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 0) load_k(kA, nboff, 1);
            mx = max4(sa, mx, (t + 1) * 64, j, MASKED);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < nt) rebase(sa, xhalf_max(mx));
        boff = nboff;
#ifdef SUPIR_ATTN_TIMELINE
        asm volatile("s_nop 0" ::"v"(o[0][0]), "v"(o[1][15]), "v"(sa[0]));
        {
            ATL(tl_5);
            tl_sync += tl_1 - tl_0;
            tl_a0 += tl_2 - tl_1;
            tl_b0 += tl_3 - tl_2;
            tl_a1 += tl_4 - tl_3;
            tl_b1 += tl_5 - tl_4;
        }
#endif
    };
    const int n_plain = p.causal ? 0 : (tail_mask ? (nt - 2 > 0 ? nt - 2 : 0) : nt);
    int t = 0;
    for (; t < n_plain; ++t) tile(t, attn_flag<false>{});
    for (; t < nt; ++t) tile(t, attn_flag<true>{});
    ATL(tl_loop1);
    attn_wait_vmcnt<0>();   // (S > 3 only: reloads issued by the last iterations still target this workgroup's LDS)

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if constexpr (!R4) {
        if (q_ok) {
            bf16_t* Op = p.O + ((size_t)b * p.Tq + q) * p.ldo + h * 64;
#pragma unroll
            for (int df = 0; df < 2; ++df)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    u16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[df][rg * 4 + e] * inv);
                    supir_store8(Op + df * 32 + 8 * rg + 4 * half, __builtin_bit_cast(u32x2, ov));
                }
        }
    } else {
        // O^T fragments -> this wave's 32 x 64 bf16 block, staged through LDS (row stride 144 B: conflict-free 8-byte writes and
        // 16-byte reads), then stored as whole 128-byte rows: 8 lanes x 16 B cover one query's head slice, 4 store instructions per
        // lane instead of 8 eight-byte stores scattered over 32 rows each (the store tail was issue-bound: 3.4 k cycles per wave at
        // Tk = 77, half of the kernel's time there).  The ring is free: every wave has passed its last tile's reads (barrier).
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr int ORS = 144;
        char* o_stage = smem + wave * (32 * ORS);
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                u16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[df][rg * 4 + e] * inv);
                *(u16x4*)(o_stage + l31 * ORS + (df * 32 + 8 * rg + 4 * half) * 2) = ov;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private block: no barrier needed
        {
            const int q0 = qb * QB + wave * 32;
            bf16_t* Ob = p.O + ((size_t)b * p.Tq + q0) * p.ldo + h * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                if (q0 + row < p.Tq) {
                    const f32x4 piece = *(const f32x4*)(o_stage + row * ORS + ch * 16);
                    supir_store16(Ob + (size_t)row * p.ldo + ch * 8, piece);
                }
            }
        }
    }
#ifdef SUPIR_ATTN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g_atl_buf && lane == 0) {
        ATL(tl_end);
        unsigned long long* ob = g_atl_buf + ((size_t)blockIdx.x * NW + wave) * 8;
        ob[0] = tl_loop1 - tl_t0;   // prologue + loop
        ob[1] = tl_sync;
        ob[2] = tl_a0;
        ob[3] = tl_b0;
        ob[4] = tl_a1;
        ob[5] = tl_b1;
        ob[6] = tl_end - tl_loop1;
        ob[7] = tl_end - tl_t0;
    }
#endif
}

static int attn_check(const AttnArgs& a) {
    if (a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tk <= 0) return SUPIR_ERR_ARG;
    if ((a.ldq | a.ldk | a.ldvt) % 8 != 0 || a.ldo % 4 != 0) return SUPIR_ERR_SHAPE;
    if (a.ldvt < ((a.Tk + 63) / 64) * 64) return SUPIR_ERR_SHAPE;
    return SUPIR_OK;
}

int supir_attn_launch(const AttnArgs& a, hipStream_t st) {
    const int rc = attn_check(a);
    if (rc != SUPIR_OK) return rc;
    // ring depth 3 (4 measured equal or slower), 4 waves = 128 queries per workgroup (2 waves measured slower on every shape),
    // softmax scale folded into Q (6-10 % faster than a multiply per element; network-level parity unchanged:
    // profiles/r02/attn_pipelined_network_parity_and_step.log)
    AttnArgsN<1> pp;
    pp.p[0] = a;
    const dim3 grid(((a.Tq + 127) / 128) * a.H * a.B);
    const bool aligned = a.ldo % 8 == 0 && (((size_t)a.O) & 15) == 0;
    const int knob4 = supir_debug_knob_value(3);  // tools only: 1 = round-3 kernel, 2 = always eight waves, 3 = always four waves (round-4 form),
    const bool ds = knob4 != 4;                   //             4 = the round-4 policy with the fp32 row sum of rounds 1-3 (DS = false)
    const int knob = knob4 == 4 ? 0 : knob4;
    // EIGHT waves = 256 query rows per workgroup (two waves per SIMD on one K / V^T ring: half the global -> LDS instructions and
    // bytes per wave, the two waves of a SIMD fill each other's issue gaps) where the whole launch is ONE round of such workgroups
    // (<= 256: one per CU): (B2, H20, 1024^2) = 160 workgroups: 21.1 -> 19.8 us.  With more than one round the coarser grid loses
    // ((B2, H10, 4096^2) = 320 workgroups of which a CU holds one: 100 -> 128 us): four waves there.  Both forms are bitwise equal
    // (profiles/r04/micro_flash_attention_*.log).
    const int nwg8 = ((a.Tq + 255) / 256) * a.H * a.B;
    const bool eight = aligned && (knob == 2 || (knob == 0 && nwg8 <= 256 && a.Tq >= 256 && a.Tk >= 256 && !a.causal));
    if (eight) {
        if (ds) SUPIR_LAUNCH((attn_d64_pipe_kernel<3, 8, true, 1, true, true>), dim3(nwg8), dim3(512), 0, st, pp);
        else SUPIR_LAUNCH((attn_d64_pipe_kernel<3, 8, true, 1, true, false>), dim3(nwg8), dim3(512), 0, st, pp);
        return SUPIR_LAUNCH_STATUS();
    }
    // the round-4 epilogue needs 16-byte aligned output rows; anything else (and knob 3 = 1) runs the round-3 form
    if (knob == 1 || !aligned) {
        SUPIR_LAUNCH((attn_d64_pipe_kernel<3, 4, true, 1, false, true>), grid, dim3(256), 0, st, pp);
    } else if (ds) {
        SUPIR_LAUNCH((attn_d64_pipe_kernel<3, 4, true, 1, true, true>), grid, dim3(256), 0, st, pp);
    } else {
        SUPIR_LAUNCH((attn_d64_pipe_kernel<3, 4, true, 1, true, false>), grid, dim3(256), 0, st, pp);
    }
    return SUPIR_LAUNCH_STATUS();
}

// two problems with the same grid (B, H, Tq) in one launch; Tk, strides and the causal flag are read per problem
int supir_attn_launch_n(const AttnArgs* a, int n, hipStream_t st) {
    if (n == 1) return supir_attn_launch(a[0], st);
    if (n != 2) return SUPIR_ERR_SHAPE;
    AttnArgsN<2> pp;
    for (int q = 0; q < 2; ++q) {
        const int rc = attn_check(a[q]);
        if (rc != SUPIR_OK) return rc;
        pp.p[q] = a[q];
    }
    if (a[0].B != a[1].B || a[0].H != a[1].H || a[0].Tq != a[1].Tq) return SUPIR_ERR_SHAPE;
    const int nwg = ((a[0].Tq + 127) / 128) * a[0].H * a[0].B;
    SUPIR_LAUNCH((attn_d64_pipe_kernel<3, 4, true, 2, false, true>), dim3(8 * ((nwg + 3) / 4)), dim3(256), 0, st, pp);
    return SUPIR_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// Row softmax for the VAE mid-block single-head attention (head dim 512, reference:
// sgm/modules/diffusionmodules/model.py:177-192, 228-256). With 288 GB of HBM the [T][T] score matrix is simply
// materialised in fp32 by the GEMM kernel (out_mode 1); this kernel turns each fp32 row into bf16 probabilities.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, bf16_t* __restrict__ P,
                                                            int T, int Tpad, long lds_, long ldp, float scale) {
    // columns [0, T) are the keys; [T, Tpad) is K-padding of the following P.V GEMM and is written as exact zeros
    __shared__ float red[8];
    const float* s = S + (size_t)blockIdx.x * lds_;
    bf16_t* pr = P + (size_t)blockIdx.x * ldp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid * 4; i < T; i += 1024) {
        const f32x4 v = *(const f32x4*)(s + i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i + e < T) mx = fmaxf(mx, v[e]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid * 4; i < T; i += 1024) {
        const f32x4 v = *(const f32x4*)(s + i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i + e < T) sum += __expf((v[e] - mx) * scale);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int i = tid * 4; i < Tpad; i += 1024) {
        const f32x4 v = *(const f32x4*)(s + i);
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (i + e < T) ? f2bf(__expf((v[e] - mx) * scale) * inv) : (u16)0;
        *(u16x4*)(pr + i) = o;
    }
}

int supir_softmax_rows_launch(const float* S, bf16_t* P, int rows, int T, int Tpad, long lds_, long ldp, float scale,
                              hipStream_t st) {
    if (rows <= 0 || T <= 0 || Tpad < T || Tpad % 4 != 0 || lds_ % 4 != 0 || ldp % 4 != 0 || lds_ < Tpad || ldp < Tpad)
        return SUPIR_ERR_SHAPE;
    SUPIR_LAUNCH(softmax_rows_kernel, dim3(rows), dim3(256), 0, st, S, P, T, Tpad, lds_, ldp, scale);
    return SUPIR_LAUNCH_STATUS();
}
