// Flash attention forward for ONE head of dimension 512: the VAE mid-block self attention
// (sgm/modules/diffusionmodules/model.py:177-192 AttnBlock.attention == :228-256 MemoryEfficientAttnBlock.attention;
// SUPIR/utils/tilevae.py:276,335 call the same through xformers).  softmax(Q K^T / sqrt(512)) V, no mask, no dropout.
//
// Before this kernel the score matrix was materialised: at 1024 x 1024 px the mid block has T = 16 384 tokens, i.e. a
// 1 GiB fp32 [T][T] matrix written by one GEMM, read and re-written as 0.5 GiB of probabilities by a softmax kernel and read
// again by a second GEMM (~3 GB of HBM traffic per call, four calls per image); here nothing but Q, K, V^T and O touches HBM.
//
// Layout:  Q [B][Tq][ldq], K [B][Tk][ldk] (512 contiguous channels per token), Vt [B][512][ldvt] = V transposed per batch
// (the SUPIR_OUT_BF16_T output of the v projection; ldvt >= round_up(Tk, 32), padding finite), O [B][Tq][ldo].
//
// One workgroup = NW waves = 32*NW query rows; a wave owns 32 queries and the whole 32 x 512 output tile: 16 accumulator
// blocks of v_mfma_f32_32x32x16 = 256 registers (AGPRs), the Q fragments another 128 VGPRs -- one wave per SIMD, by design:
// with head dim 512 the register file, not LDS, is what bounds the query tile.  Keys are consumed 32 at a time: K tile
// [32][512] (32 KB) and V^T tile [512][32] (32 KB), double buffered through LDS by global_load_lds (128 KB).
//   S^T = K.Q^T  (32 MFMAs, two independent accumulator chains; operands swapped so that a lane owns ONE query column:
//                 the row maximum is one v_permlane32_swap away and P needs no cross-lane exchange)
//   P   = exp2((S^T - m) * scale * log2(e));  m = the EXACT row maximum of the raw scores, found by a first pass over the K
//                 tiles alone (S^T only) -- with 256 accumulators an online rescale is what must not happen (see pass 1)
//   O^T += V^T.P  (32 MFMAs)
// K rows are read with bits 2/3 of the row index swapped, which makes the 8 P values a lane holds per 16-key block the 8
// CONSECUTIVE keys of one 16-byte chunk of a V^T row (see attention.hip).  Both tiles are stored XOR-swizzled (the swizzle is
// applied on the global-source side of the LDS DMA) so that every ds_read_b128 lane group hits 16 distinct 16-byte slots.
#include "kernels.h"

namespace {

template <int N>
__device__ __forceinline__ void d512_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float d512_xhalf_max(float x) {   // max over lanes l and l ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

struct Attn512Args {
    const bf16_t* Q;
    const bf16_t* K;
    const bf16_t* Vt;
    bf16_t* O;
    int B, Tq, Tk;
    int ldq, ldk, ldvt, ldo;
    float scale_log2e;
    // key-split form (SPLIT): workgroup (b, qb, sp) attends to key tiles [sp * tps, min(nt, (sp + 1) * tps)) only and leaves its
    // normalised fp32 output and (row maximum, row sum) in the workspace; attn_d512_combine_kernel merges the splits
    int nsplit, tps;
    float* part;   // [nsplit][B][Tq][512] fp32
    float* ml;     // [nsplit][B][Tq][2]  (m in raw score units, l)
};

constexpr int KT = 32;              // keys per tile
constexpr int STAGE_BYTES = 65536;  // K tile (32 KB) + V^T tile (32 KB)

template <int NW, bool SPLIT>
__global__ __launch_bounds__(64 * NW) void attn_d512_kernel(const Attn512Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // two stages
    constexpr int QB = 32 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    const int nqb = (p.Tq + QB - 1) / QB;
    // SPLIT: the splits of one query block are neighbouring block ids (different XCDs); the query blocks that share a key range follow
    // each other at stride nsplit, i.e. on the same XCD whenever nsplit divides 8
    const int sp = SPLIT ? (int)(blockIdx.x % (unsigned)p.nsplit) : 0;
    const int bq = SPLIT ? (int)(blockIdx.x / (unsigned)p.nsplit) : (int)blockIdx.x;
    const int b = bq / nqb, qb = bq - b * nqb;
    const char* Kb = (const char*)(p.K + (size_t)b * p.Tk * p.ldk);
    const char* Vb = (const char*)(p.Vt + (size_t)b * 512 * p.ldvt);

    const int q = qb * QB + wave * 32 + l31;
    const bool q_ok = q < p.Tq;
    const int qc = q_ok ? q : p.Tq - 1;
    const bf16_t* Qp = p.Q + ((size_t)b * p.Tq + qc) * p.ldq + 8 * half;
    const float c = p.scale_log2e;
    const int nt = (p.Tk + KT - 1) / KT;
    // this workgroup's key tiles [t_lo, t_hi) (never empty: the launcher sizes nsplit so that (nsplit - 1) * tps < nt)
    const int t_lo = SPLIT ? sp * p.tps : 0;
    const int t_hi = SPLIT ? (t_lo + p.tps < nt ? t_lo + p.tps : nt) : nt;

    // ---- loader: one wave-instruction moves 1 KB.  K: instruction r = key row r of the tile (64 chunks of 16 B); the lane
    // that fills LDS chunk position `lane` fetches logical chunk lane ^ (r & 15).  V^T: instruction i = channel rows
    // 16 i .. 16 i + 15 (4 chunks each); position (row, lane & 3) fetches logical chunk (lane & 3) ^ ((row >> 2) & 3).
    // Source address = wave-uniform 64-bit base (SGPR pair) + a per-lane UNSIGNED 32-bit byte offset, so that the loads take the
    // scalar-base form and no 64-bit per-lane address is ever formed (the first version spilled four of them and reloaded them from
    // scratch between the loads: every reload's vmcnt(0) also waited for the LDS-DMA issued before it).
    constexpr int LPT = 32 / NW;   // wave-instructions per wave for each of the two tiles
    const unsigned lane16 = (unsigned)lane << 4;
    const unsigned voff = (unsigned)(lane >> 2) * (unsigned)p.ldvt * 2u + ((unsigned)((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    auto stage_k = [&](int t, int soff, int i) {   // i in [0, LPT); K tile t -> smem + soff
        const int row = i * NW + wave;
        int kr = t * KT + row;
        kr = kr < p.Tk ? kr : p.Tk - 1;   // ragged last tile / tiles past the end: re-read the last valid key (masked or unused)
        const char* base = Kb + (size_t)kr * p.ldk * 2;
        glds16(base + (lane16 ^ ((unsigned)(row & 15) << 4)), smem + soff + row * 1024);
    };
    auto stage_v = [&](int t, int soff, int i) {   // rows 16 ins .. 16 ins + 15 of V^T; (row >> 2) & 3 == (lane >> 4) & 3 for all of them
        const int ins = i * NW + wave;
        const char* base = Vb + (size_t)ins * 16 * p.ldvt * 2 + (size_t)t * (KT * 2);
        glds16(base + voff, smem + soff + 32768 + ins * 1024);
    };

    // ---- LDS fragment addresses (bytes inside a stage)
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // MFMA row i of S^T holds key swap_bits(2,3)(i)
    const int kx = (krow & 15) ^ half;      // chunk 2*ks + half lives at position ((2*ks) & 48) | ((((2*ks) & 15)) ^ kx)
    const int kbase = krow * 1024;
    const int vx = ((l31 >> 2) & 3) ^ half;  // chunk 2*j + half of V^T row d lives at position (2*j) ^ vx
    const int vbase = 32768 + l31 * 64;

    f32x16 o[16];
    float nm = 0.f;   // -m of this lane's query (raw score units), set by pass 1
#pragma unroll
    for (int db = 0; db < 16; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float l_run = 0.f;

    // Q fragments as stored.  The softmax scale is applied to the fp32 scores inside the exponential (16 multiplies per key tile
    // against 64 MFMAs): folding it into Q as attention.hip does costs one more 16-bit rounding of Q, which at T = 16 384 keys
    // and logits of a few nats was the largest error term of the kernel (measured 4.1e-3 rel-L2 against fp32 SDPA).
    bf16x8 qf[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) qf[ks] = *(const bf16x8*)(Qp + 16 * ks);
    // pass-1 prologue: the 128 KB of LDS are a FOUR-deep ring of K tiles here (no V^T yet): tiles 0..2 in flight
    constexpr int K_SLOT = 32768;
#pragma unroll
    for (int t0 = 0; t0 < 3; ++t0)
#pragma unroll
        for (int i = 0; i < LPT; ++i) stage_k(t_lo + t0, t0 * K_SLOT, i);

    const int klast = p.Tk - 1;
    // S^T = K.Q^T + c0 (c0 = -m of this lane's query, or 0) for the tile in stage `sb`, keys past the end -> -inf.  Eight groups of four k steps: the fragments of
    // group g+1 are read before the MFMAs of group g are issued (explicit double buffer: left to itself the compiler reads,
    // waits and multiplies one fragment at a time through a single register quad), two independent accumulator chains, and
    // `between(g)` -- where the caller issues the loader's global_load_lds for a LATER tile -- called once per group.
    auto scores = [&](const char* sb, float c0, int t, auto&& between) {
        // Both chains start from the instruction's inline-constant 0 and "- m" is added afterwards (16 VALU adds per tile).  As the
        // C operand of the first MFMA -- attention.hip's trick -- the 16-register splat of m is loop invariant, gets hoisted, does not
        // fit next to 256 accumulators + 128 Q registers and comes back as a per-tile SCRATCH reload, whose vmcnt(0) then also
        // waits for the LDS-DMA loads issued just before it.
        f32x16 s0, s1;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bf16x8 kf[2][4];
        auto ldk = [&](bf16x8 (&k4)[4], int g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ks = 4 * g + i;
                const int pos = ((2 * ks) & 48) | (((2 * ks) & 15) ^ kx);
                k4[i] = *(const bf16x8*)(sb + kbase + pos * 16);
            }
        };
        ldk(kf[0], 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) ldk(kf[(g + 1) & 1], g + 1);
            between(g);
            s0 = SUPIR_MFMA_32x32x16(kf[g & 1][0], qf[4 * g + 0], g == 0 ? zero : s0, 0, 0, 0);
            s1 = SUPIR_MFMA_32x32x16(kf[g & 1][1], qf[4 * g + 1], g == 0 ? zero : s1, 0, 0, 0);
            s0 = SUPIR_MFMA_32x32x16(kf[g & 1][2], qf[4 * g + 2], s0, 0, 0, 0);
            s1 = SUPIR_MFMA_32x32x16(kf[g & 1][3], qf[4 * g + 3], s1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (s0[r] + s1[r]) + c0;
        if (t == nt - 1) {   // the only tile that can hold keys past the end
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * KT + 16 * (r >> 3) + 8 * half + (r & 7) > klast) s[r] = -INFINITY;
        }
        return s;
    };

    // ================= pass 1: exact row maxima (K tiles only) =================
    // An online softmax would have to rescale the 256 accumulators whenever the running maximum moves; as a data-dependent
    // branch that update drags every accumulator into VGPRs at the join (measured: 1 235 spills, Q fragments in scratch).  With
    // head dim 512 the score product is a third of the work per key tile, so the maxima are computed first -- K streams through
    // LDS once more, from L2 -- and the main pass runs with a fixed m: no rescale, P <= 1, l >= 1, no overflow cases.
    {
        float mx = -INFINITY;
        for (int t = t_lo; t < t_hi; ++t) {
            const int u = t - t_lo;   // position in the ring
            // loads complete in issue order: all but the two newest tiles' have landed, i.e. Q and tile t.  The first version
            // of this kernel issued a tile's loads DURING the previous tile and then waited for vmcnt(0): the full L2 latency
            // sat on the critical path of every tile (measured 10.6 k cycles per tile against 3 k of MFMA work).
            d512_wait_vmcnt<2 * LPT>();
            __builtin_amdgcn_s_barrier();   // ... for every wave; and every wave is done with tile t-1, whose slot takes tile t+3
            asm volatile("" ::: "memory");
            const int nslot = ((u + 3) & 3) * K_SLOT;
            const f32x16 s = scores(smem + (u & 3) * K_SLOT, nm, t, [&](int g) {
                if (g < 4) {   // past the end the loader re-reads the last tile into a slot nobody reads: the count stays uniform
#pragma unroll
                    for (int i = g * LPT / 4; i < (g + 1) * LPT / 4; ++i) stage_k(t + 3, nslot, i);
                }
            });
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
        }
        nm = -d512_xhalf_max(mx);   // finite: the first key of the range is visible to every query
    }
    d512_wait_vmcnt<0>();           // the ring's trailing (unused) loads still target this workgroup's LDS
    __builtin_amdgcn_s_barrier();   // every wave is done with the last K tile before stage 0 is refilled
    asm volatile("" ::: "memory");

    // ================= pass 2: P = exp2((S^T - m) c), O^T += V^T.P =================
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        stage_k(t_lo, 0, i);
        stage_v(t_lo, 0, i);
    }
    for (int t = t_lo; t < t_hi; ++t) {
        const int u = t - t_lo;
        d512_wait_vmcnt<0>();              // this wave's share of tile t has landed ...
        __builtin_amdgcn_s_barrier();      // ... and so has everybody else's; all waves are done with tile t-1
        asm volatile("" ::: "memory");
        const char* sb = smem + (u & 1) * STAGE_BYTES;
        const int noff = ((u + 1) & 1) * STAGE_BYTES;
        const bool more = t + 1 < t_hi;
        // tile t+1: all 2 * LPT loads on the first four score groups, so that they have the rest of this tile to land
        const f32x16 s = scores(sb, nm, t, [&](int g) {
            if (more && g < 4) {
#pragma unroll
                for (int i = (g & 1) * LPT / 2; i < ((g & 1) + 1) * LPT / 2; ++i) {
                    if (g < 2)
                        stage_k(t + 1, noff, i);
                    else
                        stage_v(t + 1, noff, i);
                }
            }
        });
        // V^T fragments of the first pair of channel blocks are requested before the exponentials
        bf16x8 va[2][4];
        auto ldv = [&](bf16x8 (&v4)[4], int g) {
#pragma unroll
            for (int dbl = 0; dbl < 2; ++dbl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    v4[dbl * 2 + j] = *(const bf16x8*)(sb + vbase + (2 * g + dbl) * 2048 + (((2 * j) ^ vx) << 4));
        };
        ldv(va[0], 0);
        bf16x8 pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(s[r] * c);   // s = q.k - max_k q.k <= 0 in raw units; c = scale * log2(e)
            l_run += pv;
            pf[r >> 3][r & 7] = (bf16_t)pv;
        }
        __builtin_amdgcn_sched_barrier(0);
        // eight groups of two channel blocks: two accumulator chains alternate, the next group's fragments are in flight
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) ldv(va[(g + 1) & 1], g + 1);
            o[2 * g] = SUPIR_MFMA_32x32x16(va[g & 1][0], pf[0], o[2 * g], 0, 0, 0);
            o[2 * g + 1] = SUPIR_MFMA_32x32x16(va[g & 1][2], pf[0], o[2 * g + 1], 0, 0, 0);
            o[2 * g] = SUPIR_MFMA_32x32x16(va[g & 1][1], pf[1], o[2 * g], 0, 0, 0);
            o[2 * g + 1] = SUPIR_MFMA_32x32x16(va[g & 1][3], pf[1], o[2 * g + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if constexpr (SPLIT) {
        if (q_ok) {
            const size_t row = ((size_t)sp * p.B + b) * p.Tq + q;
            float* Pp = p.part + row * 512;
#pragma unroll
            for (int db = 0; db < 16; ++db)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = o[db][rg * 4 + e] * inv;
                    supir_store16(Pp + db * 32 + 8 * rg + 4 * half, ov);
                }
            if (half == 0) {
                p.ml[row * 2] = -nm;
                p.ml[row * 2 + 1] = l_tot;
            }
        }
        return;
    }
    if (q_ok) {
        bf16_t* Op = p.O + ((size_t)b * p.Tq + q) * p.ldo;
#pragma unroll
        for (int db = 0; db < 16; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                u16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[db][rg * 4 + e] * inv);
                supir_store8(Op + db * 32 + 8 * rg + 4 * half, __builtin_bit_cast(u32x2, ov));
            }
    }
}


// Merge of the key splits (fixed order, reproducible): out = sum_h w_h O_h,  w_h = l_h 2^((m_h - m) c) / sum_h' l_h' 2^((m_h' - m) c),
// m = max_h m_h.  One thread per (query row, 4 channels); a row's 1 KB of bf16 output is one contiguous store per 128 threads.
__global__ __launch_bounds__(256) void attn_d512_combine_kernel(const float* __restrict__ part, const float* __restrict__ ml,
                                                                bf16_t* __restrict__ O, long rows, int ldo, int nsplit, float c) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = gid >> 7;
    const int cg = (int)(gid & 127);
    if (row >= rows) return;
    float m = -INFINITY;
    for (int h = 0; h < nsplit; ++h) m = fmaxf(m, ml[((size_t)h * rows + row) * 2]);
    float den = 0.f;
    for (int h = 0; h < nsplit; ++h) {
        const float* st = ml + ((size_t)h * rows + row) * 2;
        den += st[1] * __builtin_amdgcn_exp2f((st[0] - m) * c);
    }
    const float inv = 1.0f / den;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < nsplit; ++h) {
        const float* st = ml + ((size_t)h * rows + row) * 2;
        const float w = st[1] * __builtin_amdgcn_exp2f((st[0] - m) * c) * inv;
        const f32x4 v = *(const f32x4*)(part + ((size_t)h * rows + row) * 512 + cg * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += w * v[e];
    }
    u16x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = f2bf(acc[e]);
    supir_store8(O + (size_t)row * ldo + cg * 4, __builtin_bit_cast(u32x2, ov));
}

}  // namespace

constexpr int D512_NW = 4;   // 128 queries per workgroup: half the K / V^T stream (and LDS-DMA issue) per FLOP of the 2-wave form

// Key splits.  A launch has B * ceil(Tq / 128) workgroups of one wave per SIMD, one per CU (128 KB of LDS): 128 at T = 16 384 (the mid
// block of a 1024^2 image), 32 at T = 4096 -- half / an eighth of the 256 CUs -- and 328 (1.28 rounds = 2) for eight stacked 5184-token
// tiled-VAE tiles.  With the keys split over `ns` workgroup sets the same work is cut finer; the merge costs 2 x ns x 2 KB per query row
// of extra traffic.  requested <= 0: the count that minimises  rounds-of-256-workgroups x key tiles per split + the merge  (in units of
// one 32-key tile of one workgroup, ~3 us: the merge is ~0.6 of that per split and 4096 rows, a split launch 2 more), among the counts
// with >= 4 tiles per split, <= 16, for >= 16 key tiles; a split must win by 3 %.  Measured (profiles/r04/micro_attn_d512_key_split_sweep.log):
// (1, 16 384): 1 518 us unsplit, 912 / 906 / 950 / 1 073 with 2 / 4 / 8 / 16; (1, 4096): 399, 209 / 122 / 90 / 125; (4, 4096): 412, 261 / 310 / 375.
static int d512_splits(int B, int Tq, int Tk, int requested) {
    const int nt = (Tk + KT - 1) / KT;
    const long nwg1 = (long)B * ((Tq + 32 * D512_NW - 1) / (32 * D512_NW));
    long ns = requested;
    if (requested <= 0) {
        ns = 1;
        if (nt >= 16) {
            const double merge = 0.6 * (double)B * Tq / 4096.0;
            double best = (double)((nwg1 + 255) / 256) * nt;
            for (int c = 2; c <= 16; ++c) {
                const int tps = (nt + c - 1) / c;
                if (tps < 4) break;
                if ((nt + tps - 1) / tps != c) continue;   // same split as a smaller count
                const double cost = (double)((nwg1 * c + 255) / 256) * tps + merge * c + 2.0;
                if (cost < 0.97 * best) {
                    best = cost;
                    ns = c;
                }
            }
        }
    }
    if (ns > 16) ns = 16;
    if (ns > nt) ns = nt;
    if (ns < 1) ns = 1;
    const int tps = (nt + (int)ns - 1) / (int)ns;
    return (nt + tps - 1) / tps;   // no empty split
}

size_t supir_attn_d512_workspace_bytes(int B, int Tq, int Tk, int splits) {
    if (B <= 0 || Tq <= 0 || Tk <= 0) return 0;
    const int ns = d512_splits(B, Tq, Tk, splits);
    return ns <= 1 ? 0 : (size_t)ns * B * Tq * (512 + 2) * sizeof(float);
}

int supir_attn_d512_launch(const bf16_t* Q, const bf16_t* K, const bf16_t* Vt, bf16_t* O, int B, int Tq, int Tk, int ldq, int ldk,
                           int ldvt, int ldo, float scale, int splits, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (B <= 0 || Tq <= 0 || Tk <= 0 || !(scale > 0.f)) return SUPIR_ERR_ARG;   // the row maxima are taken on the unscaled scores
    if ((ldq | ldk | ldvt) % 8 != 0 || ldo % 4 != 0 || ldq < 512 || ldk < 512 || ldo < 512) return SUPIR_ERR_SHAPE;
    if (ldvt < ((Tk + KT - 1) / KT) * KT) return SUPIR_ERR_SHAPE;
    constexpr int NW = D512_NW;
    const int ns = workspace ? d512_splits(B, Tq, Tk, splits) : 1;   // no workspace: the single-pass form
    if (ns > 1 && (workspace_bytes < supir_attn_d512_workspace_bytes(B, Tq, Tk, splits) || (((size_t)workspace) & 15))) return SUPIR_ERR_ARG;
    Attn512Args a{Q, K, Vt, O, B, Tq, Tk, ldq, ldk, ldvt, ldo, scale * 1.4426950408889634f, ns, 0, nullptr, nullptr};
    const long nwg = (long)B * ((Tq + 32 * NW - 1) / (32 * NW)) * ns;
    if (nwg > 0x7fffffffL) return SUPIR_ERR_SHAPE;
    static bool attr_set = false;
    if (!attr_set) {
        if (supir_note_hip_status(hipFuncSetAttribute((const void*)attn_d512_kernel<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      2 * STAGE_BYTES)) != SUPIR_OK ||
            supir_note_hip_status(hipFuncSetAttribute((const void*)attn_d512_kernel<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      2 * STAGE_BYTES)) != SUPIR_OK)
            return SUPIR_ERR_HIP;
        attr_set = true;
    }
    if (ns == 1) {
        SUPIR_LAUNCH((attn_d512_kernel<NW, false>), dim3((unsigned)nwg), dim3(64 * NW), 2 * STAGE_BYTES, st, a);
        return SUPIR_LAUNCH_STATUS();
    }
    const int nt = (Tk + KT - 1) / KT;
    const long rows = (long)B * Tq;
    a.tps = (nt + ns - 1) / ns;
    a.part = (float*)workspace;
    a.ml = a.part + (size_t)ns * rows * 512;
    SUPIR_LAUNCH((attn_d512_kernel<NW, true>), dim3((unsigned)nwg), dim3(64 * NW), 2 * STAGE_BYTES, st, a);
    if (const int rc = SUPIR_LAUNCH_STATUS()) return rc;
    const long nblk = (rows * 128 + 255) / 256;
    if (nblk > 0x7fffffffL) return SUPIR_ERR_SHAPE;
    SUPIR_LAUNCH(attn_d512_combine_kernel, dim3((unsigned)nblk), dim3(256), 0, st, a.part, a.ml, O, rows, ldo, ns, a.scale_log2e);
    return SUPIR_LAUNCH_STATUS();
}
