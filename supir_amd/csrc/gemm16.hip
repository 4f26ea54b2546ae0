// bf16 MFMA GEMM on v_mfma_f32_16x16x32_bf16 for the shapes whose tile grid can be made EXACTLY a multiple of the 256 CUs
// (tiles 32..35 of the tile table).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )      same contract as gemm.hip, for M % BM == 0, N % BN == 0, K % (64 KS) == 0
//
// Why a second kernel family.  70 % of the GEMM launches of a 1024^2 step have M = 2048 and N = 1280 (to_q / to_out / ff.net.2 /
// proj_in / proj_out of the 1280-wide SpatialTransformers, sgm/modules/attention.py:100-106,213-219,587,611).  On the
// 32x32x16 tiles of gemm.hip that is 160 tiles of 128x128 (96 of 256 CUs idle) or 640 tiles of 64x64 (2.5 waves of workgroups,
// twice the L1 traffic per FLOP); measured hot: 17.4 / 14.6 us = 385-460 TFLOP/s whatever the tile.  What bounds a CU is its
// global->LDS fill rate (64 B/clk) and the bytes it can keep in flight (its LDS): a tile needs (BM + BN) * 128 B per K step, so
// the right tile is the squarest one that gives exactly one workgroup per CU: 2048 x 1280 / 256 = 128 x 80.  80 = 5 x 16, hence
// the 16x16x32 MFMA (same FLOP rate as 32x32x16).  N = 2560 (fused to_q|to_k) -> 128 x 160; the GEGLU projection
// (2048 x 10240, attention.py:87) -> 256 x 160 = 512 workgroups = two full rounds.
//
//  tile  BM x BN    waves          K groups  ring   LDS      used for
//   32   128 x  80  4x1 per group     2       2   104 KB   M = 2048, N = 1280 (also N = 640 at M = 8192)
//   33   128 x 160  2x2 per group     2       2   144 KB   N = 2560, M = 8192 x N = 640
//   34   256 x 160  8x1               1       3   156 KB   N = 10240 (GEGLU epilogue: value / gate interleaved per 16 rows of W)
//   35   128 x  80  4x1 per group     2       3   156 KB   tile 32 with a 3-deep ring (two K steps of loads in flight per group)
//   39   256 x 128  4x2               1       3   144 KB   128-channel outputs at M >= 64 K rows (VAE convolutions, sgm/modules/
//                                                          diffusionmodules/model.py:55-148: Cout = 128 at 512^2 / 1024^2)
//   40   256 x 256  4x2               1       2   128 KB   256 / 512-channel outputs of the VAE (32 B/clk of fill per CU: the one tile
//                                                          of the family under the 38 B/clk the loops sustain); one 32-wide K slice of
//                                                          fragments in registers at a time (128 accumulator registers)
//   45   512 x 128  4x2  8-phase      1       2   160 KB   the 128-channel outputs on the same schedule: the 256 x 128 tile needs 48 B/clk of fill per
//                                                          CU for its MFMAs (fill-bound: its K step ran 3222 cycles for 1024 of MFMA on 128 -> 128
//                                                          at 1024^2), 512 x 128 needs 40
//   42   256 x 256  2x4  8-phase      1       2   128 KB   the same outputs on the eight-phase ping-pong schedule (round 5, below): half-tile
//                                                          staging with three half-tiles in flight across the barriers, the two waves of
//                                                          every SIMD alternating between a fragment-read / load-issue segment and a
//                                                          16-MFMA segment
//
//  * 512 threads.  KS = 2: two K groups of four waves, group g takes K steps g, g+2, ... from its own LDS ring, so every SIMD
//    holds two waves (one per group) whose load issue / LDS reads / MFMAs interleave; the partial accumulators are
//    reduce-scattered through LDS after the loop (group g finishes token-fragment rows [g*MI/2, (g+1)*MI/2)) and both groups
//    run the epilogue on their half.  KS = 1: eight waves on one ring;
//  * global -> LDS by global_load_lds_dwordx4, 8 rows (1 KB) per wave instruction, source-side XOR swizzle
//    (chunk ^= (row >> 1) & 7), undone by the ds_read_b128 fragment reads: conflict-free for the 16-row x 4-chunk
//    fragment of the 16x16x32 MFMA (checked per ds_read_b128 lane group); S-deep ring, ONE raw s_barrier per K step, counted
//    vmcnt for S = 3 (every wave issues the same number of loads per stage);
//  * operands swapped (a = W rows, b = A rows): a lane ends with 4 consecutive channels of one token; the fused epilogue
//    is the one of gemm.hip (bias, LayerNorm fold / row statistics, row bias, residual, SiLU, GEGLU, alpha; bf16 output staged
//    through LDS for 16-byte row-contiguous stores); the transposed (V^T) variant swaps the operands back so that a lane
//    holds 4 consecutive tokens of one channel;
//  * exact shapes only (the dispatcher refuses anything else): no bounds checks anywhere in the kernel.
#include "kernels.h"
#include <type_traits>

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void g16_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef SUPIR_G16_TIMELINE
// tools/probes/g16_timeline.py only (never defined in the product build): per-wave s_memtime stamps of the kernel's phases
__device__ unsigned long long* g16_tl_buf;
extern "C" void supir_g16_tl_set(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g16_tl_buf), &p, sizeof(p)); }
#define G16_TL(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
// in-loop stamps (round 6): only at points where lgkmcnt is 0 anyway (s_memtime returns through lgkmcnt): loop top, after the counted
// wait, after the barrier, after the step's last MFMA + lgkmcnt(0).  ~4 scalar-cache round trips per K step: the split is what counts
#define G16_TLS(var) var = __builtin_amdgcn_s_memtime()
#define G16_TLACC(acc, a, b) acc += (b) - (a)
#else
#define G16_TL(var)
#define G16_TLS(var)
#define G16_TLACC(acc, a, b)
#endif

// zero page for the halo / padding rows of the implicit-GEMM (3x3 convolution) loader
__device__ __attribute__((aligned(256))) uint32_t g16_zero_page[64];

// MIXED (fused q|k|v projection, supir_gemm_bf16_qkv): tile columns [0, n_split) take the normal epilogue into C, tile columns
// [n_split, N) the transposed one into C2 (V^T, channel index n - n_split); decided per workgroup (tile_n), both main loops
// (they differ in the MFMA operand order) are in the kernel.
//
// NP = 2 (supir_gemm_grouped): two independent problems of identical shape in one grid.  Block b runs on XCD b % 8; problem q owns XCDs
// [4 q, 4 q + 4) -- its operands stay in those four L2s -- and maps its tiles over them exactly as a single problem does over eight.
// Used for the layer pairs GLVControl and the UNet encoder execute with identical shapes on independent data
// (SUPIR/modules/SUPIR_v0.py:499-540 next to :600-625): M = 2048 tokens per problem fill only half the machine with tiles big
// enough to be fed from L2 (128 x 80 at 49 FLOP per staged byte); two problems of 128 x 160 tiles are 256 workgroups at 65.
//
// Four-wave form (tile 38 = 128 x 80, 4x1 waves, ONE K group, 3-deep ring = 78 KB, round 4): half the threads and half the LDS of
// tile 35, so that TWO workgroups fit a CU -- in the two-stream step one of GLVControl's and one of the UNet encoder's (VERDICT r03
// item 2: every other tile of this family takes 104-156 KB and a CU then runs one workgroup of one chain at a time).  Same loader,
// same fragment layout, same epilogue; the K-group exchange disappears.
//
// LDS-staged HALO form of the 3x3 convolutions (HW_ > 0 = the map width; round 6, tiles 48-50).  The implicit-GEMM loader above stages every
// input pixel of a tile nine times (once per tap: (BM + BN) x 128 B per 64-wide K step); the counter passes show the price (fetch + write
// 2.2-6.1 x the operands, profiles/pmc_traffic.json) and the L2 -> LDS fill is what bounds these tiles.  Here a tile is BM / W whole rows of
// the map; per 64-channel chunk its (rows + 2) x (W + 2) input pixels -- the halo tile, zero page for what lies outside the image -- are
// staged ONCE, and the nine taps read their token fragments out of it at shifted addresses: the K loop runs (chunk, tap), the ring stages
// carry the W tiles only.  Per chunk a 128 x 80 tile on a 32 x 32 map stages 26 + 90 KB instead of 144 + 90 KB.
//   * halo pixel P = hy (W + 2) + hx lives at P * 128 B, its 16-byte chunk c at physical slot c ^ (P & 7): the 16 lanes of a ds_read_b128
//     group read 16 CONSECUTIVE pixels starting anywhere (the tap shift) and two adjacent chunks -- slot = (P & 1) * 8 + (c ^ (P & 7)) is a
//     bijection on any such window (enumerated: tools/probes/halo_swizzle_check.py), so every tap's fragment reads are conflict-free;
//   * two halo buffers (chunk parity).  The eight waves load chunk c + 2 into the buffer chunk c leaves, 8-pixel pieces (1 KB per wave
//     instruction, wave w takes pieces w, w + 8, ...), HQ pieces per wave and iteration in the NLI iterations after the last step of chunk
//     c -- early enough that the counted wait of the W ring (S = 3: everything but the previous iteration's loads has landed) covers them
//     S - 1 iterations later, before the first step of chunk c + 2;
//   * two K groups: group g takes steps g, g + 2, ... of the (chunk, tap) sequence -- taps of one chunk alternate between the groups, both
//     read the same halo buffers; the schedule above is in ITERATIONS (one step of each group), which both groups share;
//   * a wave's number of loads now varies by iteration (0..HQ halo pieces on top of the W chunks): the counted wait is selected, wave
//     uniformly, by the count of the previous iteration.
// Everything after the main loop (K-group exchange, epilogue, GroupNorm partials, prefetch) is the code of the other forms: a tile is still
// BM consecutive rows of the [M][N] output.
// Measured (profiles/r06): fetch + write per launch falls to 1.1-1.4 x the operands; texture-addresser busy 60 % -> 37 %; TIME is equal to
// the implicit-GEMM tiles within the box noise (0.94-1.09 x) -- the in-loop stamps show the loop is not bound by the fill (counted wait: 8
// cycles per step), nor by LDS (index pipe 28 % busy, no bank conflicts in either form), nor by the matrix pipe (39 %): a step is wait 8 /
// barrier 250 / reads + loads + MFMAs 1070 cycles against 640 of MFMA.  Three re-orderings of that step were built and measured on top of this
// form and of the implicit one -- the two K groups ping-ponging across two barriers, fragments software-pipelined inside each wave, the
// groups' load / MFMA order staggered -- all bitwise equal, all 5-15 % SLOWER (docs/roundlog.md section 6); they are not in the tree.
template <int BM, int BN, int WM, int WN, int KS, int S, bool TRANS, bool CONV = false, bool MIXED = false, int NP = 1, int HW_ = 0>
__global__ __launch_bounds__(64 * WM * WN * KS, (WM * WN * KS == 8 ? 2 : 1)) void gemm16_kernel(const GemmArgsN<NP> pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NX = 8 / NP;                                               // XCDs per problem
    const int prob = NP == 1 ? 0 : (int)(blockIdx.x & 7) / NX;               // wave-uniform: a scalar offset into the kernarg segment
    const GemmArgs& p = pp.p[prob];
    const int vxcd = (int)blockIdx.x & (NX - 1), vidx = (int)blockIdx.x >> 3;   // XCD inside the problem's share, index on that XCD
    G16_TL(tl_start);
    constexpr int NW = WM * WN;                          // waves per K group
    constexpr bool PH8 = S == 8;                         // the eight-phase schedule (tile 42): S names the schedule, the ring is 2 deep
    constexpr int SD = PH8 ? 2 : S;
    constexpr bool HALO = HW_ > 0;
    static_assert(!HALO || (CONV && S != 8 && !MIXED && !TRANS && NP == 1 && BM % HW_ == 0 && HW_ % 16 == 0 && WM * WN * KS == 8), "halo form");
    constexpr int H_ROWS = HALO ? BM / HW_ : 1;          // output rows of the map per tile
    constexpr int H_RS = HW_ + 2;                        // halo row stride in pixels
    constexpr int H_PIX = (H_ROWS + 2) * H_RS;
    constexpr int H_PIECES = (H_PIX + 7) / 8;            // 8-pixel (1 KB) pieces of one halo tile
    constexpr int H_BYTES = H_PIECES * 1024;
    constexpr int H_PQ = (H_PIECES + 7) / 8;             // pieces per wave (piece id = wave + 8 q, all eight waves of the workgroup)
    constexpr int H_NLI_MAX = KS == 2 ? (S == 3 ? 3 : 4) : (S == 3 ? 7 : 8);   // iterations between "chunk c's buffer is free" and "chunk c + 2's data must be covered by the wait"
    constexpr int H_HQ = (H_PQ + H_NLI_MAX - 1) / H_NLI_MAX;                   // pieces per wave and load iteration
    constexpr int H_NLI = (H_PQ + H_HQ - 1) / H_HQ;                            // load iterations per halo tile
    static_assert(!HALO || H_HQ <= 2, "halo pieces per iteration: the counted wait has three variants");
    constexpr int A_BYTES = HALO ? 0 : BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES, RING = SD * STAGE_BYTES;
    constexpr int H_OFF = KS * RING;                     // LDS: [ring of group 0 | ring of group 1 | halo 0 | halo 1]
    constexpr int A_Q = HALO ? 0 : BM / 8 / NW;          // 8-row chunks of A per wave and K step (chunk id = wave + NW q)
    constexpr int B_CH = BN / 8, B_Q = (B_CH + NW - 1) / NW;   // chunks of W per tile / per wave (the last q may be partial)
    constexpr int LOADS = A_Q + B_Q;                     // global->LDS instructions per wave and stage
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16, MIH = MI / KS;
    constexpr int NWT = NW * KS, NTHREADS = 64 * NWT;   // waves / threads per workgroup (8 / 512; tile 38: 4 / 256)
    static_assert((NWT == 8 || (NWT == 4 && KS == 1 && !MIXED && NP == 1)) && (BM / 8) % NW == 0 && (NW & 1) == 0,
                  "512 threads (or the four-wave single-group form); A chunks divide over the waves");
    static_assert(MI >= KS && MI % KS == 0 && WTN % 16 == 0 && (KS == 1 || KS == 2) && (S == 2 || S == 3 || S == 8), "tile / wave grid");
    // eight-phase schedule: 256 token rows, 8 waves; the wave tile's MI token fragments split into two halves, its NI channel fragments
    // into NA + NB parts.  Shipped: 256 x 256 as 2 x 4 waves (128 x 64 per wave, 2 + 2 fragments: tile 42).  The same code was measured
    // as 256 x 160 (4 x 2 waves, 3 + 2) and as 256 x 320 for the GEGLU projections (4 x 2, 5 + 5): bitwise the one-barrier tiles' results,
    // 5-10 % slower than tile 34 resp. equal to tile 37 -- phases of 8-12 MFMAs are too short for two barriers each, and 216 live registers
    // spill (profiles/r05/experiment_*); those instantiations are not built
    static_assert(!PH8 || ((BM == 256 || BM == 512) && KS == 1 && !TRANS && NP == 1 && NWT == 8 && (MI & 1) == 0 && NI >= 2 && (!MIXED || !CONV)), "eight-phase schedule");
    constexpr int PH_HR = WTM / 2;                       // token rows of one wave in an A half-tile
    constexpr int PH_AI = MI / 2;                        // token fragments per A half
    constexpr int PH_NA = (NI + 1) / 2, PH_NB = NI - PH_NA;   // channel fragments per wave in W part 0 / part 1
    constexpr int PH_WA_ROWS = WN * PH_NA * 16, PH_WB_ROWS = WN * PH_NB * 16;
    constexpr int PH_WA_CH = PH_WA_ROWS / 8, PH_WB_CH = PH_WB_ROWS / 8;          // 8-row chunks per W part
    constexpr int PH_WA_Q = (PH_WA_CH + 7) / 8, PH_WB_Q = (PH_WB_CH + 7) / 8;    // global -> LDS instructions per wave (the last may repeat chunk `wave`)
    constexpr int PH_A_HALF = BM * 64;                   // bytes of one A half-tile (BM / 2 rows of 128 B): 16 KB, 32 KB for the 512-row tile
    constexpr int PH_A_Q = BM / 128;                     // global -> LDS instructions per wave and A half-tile (8-row chunks wave + 8 q)
    constexpr int PH_OFF_W = 2 * PH_A_HALF, PH_OFF_WB = PH_OFF_W + PH_WA_ROWS * 128;   // LDS map of a K-tile buffer: [A0 | A1 | WA | WB]
    constexpr int PH_INFLIGHT = PH_A_Q + PH_WB_Q + PH_A_Q;   // loads per wave behind W part 0 of the next K-tile: A0, WB, A1 of the one after
    static_assert(!PH8 || (PH_WA_CH <= 24 && PH_WB_CH <= 24 && (PH_HR == 64 || (PH_HR == 32 && BM == 256)) && PH_OFF_WB + PH_WB_ROWS * 128 == STAGE_BYTES), "half-tile map");
    static_assert(!(CONV && TRANS) && !(MIXED && (TRANS || CONV || KS != 1)), "the implicit-GEMM loader has no transposed epilogue");
    // epilogue LDS map (the rings are idle by then): [0, XCH) K-group exchange, then per-wave C staging, bias / column sums,
    // row-statistics scratch
    constexpr int XCH_HALF = KS == 2 ? NW * MIH * NI * 4 * 64 * 4 : 0;   // bytes one group sends
    constexpr int XCH = 2 * XCH_HALF;
    constexpr int C_RS = WTN * 2 + 16;                         // staged row stride (bytes): 16-byte pad against bank conflicts
    constexpr int C_STAGE = 16 * C_RS;                         // one 16-token block per wave
    constexpr int OFF_CST = XCH, OFF_BIAS = OFF_CST + NWT * C_STAGE, OFF_RED = OFF_BIAS + 2 * BN * 4;
    constexpr int OFF_GN = OFF_RED + WN * BM * 8;               // GroupNorm column partials: [wave 0..NWT-1][WTN][2] fp32
    static_assert(OFF_GN + NWT * WTN * 8 <= KS * RING + (HALO ? 2 * H_BYTES : 0) - (PH8 ? 256 : 0), "epilogue scratch must fit the LDS rings");

    // wave-uniform ids as scalars (readfirstlane): LDS destinations / branches on them stay on the scalar unit
    const int bwave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // wave inside the workgroup, 0..7
    const int kg = KS == 2 ? bwave >> 2 : 0;           // K group
    const int wave = KS == 2 ? bwave & 3 : bwave;      // wave inside the group
    const int lane = (int)threadIdx.x & 63;
    const int tid = wave * 64 + lane;                  // thread inside the group
    const int wm = wave / WN, wn = wave % WN;
    const int quad = lane >> 4, l15 = lane & 15;
    char* ring = smem + kg * RING;

    const int tiles_m = p.M / BM, tiles_n = p.N / BN;
    int tile_m, tile_n;
    if (p.gm > 0) {   // 2-D XCD grid, see gemm.hip
        const int xcd = vxcd, idx = vidx;
        const int rm = tiles_m / p.gm, rn = tiles_n / p.gn;
        const int xm = xcd / p.gn, xn = xcd - xm * p.gn;
        int lm, ln;
        if (p.order == 0) { ln = idx / rm; lm = idx - ln * rm; }
        else { lm = idx / rn; ln = idx - lm * rn; }
        tile_m = xm * rm + lm;
        tile_n = xn * rn + ln;
    } else {
        // 1-D ranges: every XCD of the problem's share gets a contiguous run of tile ids (grouped launches: tiles % NX == 0)
        const int id = NP == 1 ? xcd_remap(blockIdx.x, tiles_m * tiles_n) : vxcd * (tiles_m * tiles_n / NX) + vidx;
        if (p.order == 0) { tile_n = id / tiles_m; tile_m = id - tile_n * tiles_m; }
        else { tile_m = id / tiles_n; tile_n = id - tile_m * tiles_n; }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bool tr = MIXED ? (n0 >= p.n_split) : TRANS;   // this workgroup's epilogue (wave-uniform)

    // ---- loader: wave w stages the 8-row chunks w, w+NW, w+2NW, ... of A and of W; lane -> row lane>>3, physical 16-B chunk lane&7.
    // NW is even, so the chunk ids of one wave all have the parity of w and the swizzle ((row >> 1) & 7 with row = 8*chunk + lane>>3)
    // is fixed per lane
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const bf16_t* a_src = p.A + (size_t)(m0 + wave * 8 + lrow) * p.lda + lchunk * 8 + kg * 64;
    const bf16_t* b_src = p.Wt + (size_t)(n0 + wave * 8 + lrow) * p.K + lchunk * 8 + (HALO ? 0 : kg * 64);
    const size_t a_qstride = (size_t)(8 * NW) * p.lda, b_qstride = (size_t)(8 * NW) * p.K;
    // 3x3 convolution as implicit GEMM (same scheme as gemm.hip): K runs (ky, kx, cin) and a 64-wide K step never straddles a tap
    // (Cin % 64 == 0; Cin % 128 == 0 with two K groups); the gather address of each of this lane's A rows is computed once per
    // tap; halo / padding rows read the zero page; stride 2, asymmetric padding and nearest-2x upsampling fold into the gather
    int c_iy0[CONV && !HALO ? A_Q : 1], c_ix0[CONV && !HALO ? A_Q : 1];
    const bf16_t* c_base[CONV && !HALO ? A_Q : 1];
    const bf16_t* c_tap[CONV && !HALO ? A_Q : 1];
    int c_cin0 = kg * 64, c_ky = 0, c_kx = 0;
    const int VH = p.up ? 2 * p.H : p.H, VW = p.up ? 2 * p.W : p.W;
    // eight-phase schedule: a tile's 256 rows lie in ONE batch element (dispatcher), so the image base is wave-uniform and a lane keeps a
    // 32-bit element offset per row (< 0: halo / padding) instead of two pointers -- 10 VGPRs less in a kernel that sits at the 256 limit
    int ph8_toff[CONV && PH8 ? A_Q : 1];
    const bf16_t* ph8_cbase = nullptr;
    auto conv_set_tap = [&]() {
        if constexpr (CONV && !HALO) {
#pragma unroll
            for (int q = 0; q < A_Q; ++q) {
                int iy = c_iy0[q] + c_ky, ix = c_ix0[q] + c_kx;
                const bool ok = (unsigned)iy < (unsigned)VH && (unsigned)ix < (unsigned)VW;
                if (p.up) { iy >>= 1; ix >>= 1; }
                if constexpr (PH8) ph8_toff[q] = ok ? (iy * p.W + ix) * p.lda : -1;
                else c_tap[q] = ok ? c_base[q] + ((size_t)iy * p.W + ix) * p.lda : nullptr;
            }
        }
    };
    if constexpr (CONV && !HALO) {
#pragma unroll
        for (int q = 0; q < A_Q; ++q) {
            // eight-phase schedule: q = 2 h + qq is this lane's row of A-half h (token fragments [4h, 4h + 4) of both wave rows), wave row qq
            const int m = PH8 ? m0 + (q % PH_A_Q) * 128 + (q / PH_A_Q) * PH_HR + (PH_HR == 64 ? wave * 8 : (wave >> 2) * WTM + (wave & 3) * 8) + lrow
                              : m0 + (wave + NW * q) * 8 + lrow;
            const int b = m / p.rows_per_batch, r = m - b * p.rows_per_batch;
            const int oy = r / p.OW, ox = r - oy * p.OW;
            c_iy0[q] = oy * p.stride - p.pad_t;
            c_ix0[q] = ox * p.stride - p.pad_l;
            if constexpr (!PH8) c_base[q] = p.A + (size_t)b * p.H * p.W * p.lda + lchunk * 8;
        }
        if constexpr (PH8) ph8_cbase = p.A + (size_t)(m0 / p.rows_per_batch) * p.H * p.W * p.lda + lchunk * 8;
        conv_set_tap();
    }
    // ---- halo form: scalar (wave-uniform) state.  Step t of the (chunk, tap) sequence has chunk t / 9, tap t % 9, weight K offset
    // tap * Cin + chunk * 64; group g walks t = g, g + KS, ...  h_s*: the step whose W tile is staged next; h_r*: the step being read
    int h_stap = kg, h_schunk = 0, h_rtap = kg, h_rchunk = 0;
    int h_it = 0;                   // iteration (one step of each group) about to run
    int h_next = 2;                 // next chunk whose halo tile is loaded in the loop (0 and 1: prologue)
    int h_it0 = 8 / KS + 1;         // first load iteration of chunk h_next: the iteration after the last step of chunk h_next - 2, (9 c + 8) / KS + 1
    int h_cnt_prev = 0;             // halo pieces this wave issued in the previous iteration (the counted wait depends on it)
    const int h_nchunks = HALO ? p.Cin >> 6 : 0;
    int h_poff[HALO ? H_PQ : 1];    // per lane: element offset of its 16 bytes of piece (bwave + 8 q) inside the image, < 0: zero page
    int h_p0[HALO ? MI : 1];        // per lane: halo pixel index of token l15 of fragment i at tap (0, 0)
    const bf16_t* h_img = nullptr;
    if constexpr (HALO) {
        const int b = m0 / p.rows_per_batch, y0 = (m0 - b * p.rows_per_batch) / HW_;
        h_img = p.A + (size_t)b * p.H * p.W * p.lda;
        const int lg = (lane & 7) ^ (lane >> 3);     // logical chunk of this lane's slot: pixel P = 8 piece + (lane >> 3), so P & 7 = lane >> 3
#pragma unroll
        for (int q = 0; q < H_PQ; ++q) {
            const int P = (bwave + 8 * q) * 8 + (lane >> 3);
            const int hy = P / H_RS, hx = P - hy * H_RS;
            const int y = y0 - 1 + hy, x = hx - 1;
            const bool ok = P < H_PIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)HW_;
            h_poff[q] = ok ? (y * HW_ + x) * p.lda + lg * 8 : -1;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = wm * WTM + i * 16;         // first token of the fragment inside the tile: 16 | W, so a fragment stays in one map row
            h_p0[i] = (r / HW_) * H_RS + (r % HW_) + l15;
        }
    }
    // pieces [q0, q1) of this wave's H_PQ (piece id bwave + 8 q), channel chunk `chunk`, into halo buffer chunk & 1; returns how many were
    // issued (wave-uniform).  Called with compile-time q0 / q1 only (fully unrolled), so h_poff stays in registers
    auto halo_issue = [&](int chunk, int q0, int q1) {
        int n = 0;
        if constexpr (HALO) {
            char* dst = smem + H_OFF + (chunk & 1) * H_BYTES + bwave * 1024;
            const bf16_t* base = h_img + chunk * 64;
#pragma unroll
            for (int q = 0; q < H_PQ; ++q) {
                if (q >= q0 && q < q1 && bwave + 8 * q < H_PIECES) {
                    const bf16_t* src = h_poff[q] >= 0 ? base + h_poff[q] : (const bf16_t*)g16_zero_page;
                    glds16(src, dst + q * 8192);
                    ++n;
                }
            }
        }
        return n;
    };
    // one global->LDS instruction: q < A_Q -> A chunk wave + NW q, else W chunk wave + NW (q - A_Q).  A wave whose last W chunk
    // would fall beyond the tile skips it when the ring is drained with vmcnt(0) (S == 2); with counted waits (S == 3) it re-loads
    // its first W chunk instead (same bytes to the same place), so that every wave has the same number of loads in flight
    auto stage_one = [&](int buf, int q) {
        char* sA = ring + buf * STAGE_BYTES;
        if (q < A_Q) {
            if constexpr (CONV) {
                const bf16_t* src = c_tap[q] ? c_tap[q] + c_cin0 : (const bf16_t*)g16_zero_page;
                glds16(src, sA + (wave + NW * q) * 1024);
            } else {
                glds16(a_src + q * a_qstride, sA + (wave + NW * q) * 1024);
            }
        } else {
            const int qb = q - A_Q;
            const bf16_t* bs = HALO ? b_src + (h_stap * p.Cin + h_schunk * 64) : b_src;
            if ((B_CH % NW == 0) || qb < B_Q - 1 || wave < (B_CH % NW))
                glds16(bs + qb * b_qstride, sA + A_BYTES + (wave + NW * qb) * 1024);
            else if (S > 2)
                glds16(bs, sA + A_BYTES + wave * 1024);
        }
    };
    auto stage_advance = [&]() {
        if constexpr (HALO) {
            h_stap += KS;
            if (h_stap >= 9) { h_stap -= 9; ++h_schunk; }
            return;
        }
        a_src += 64 * KS;
        b_src += 64 * KS;
        if constexpr (CONV) {
            c_cin0 += 64 * KS;
            if (c_cin0 >= p.Cin) {   // next tap (Cin % (64 KS) == 0: the K groups stay tap-aligned)
                c_cin0 -= p.Cin;
                if (++c_kx == 3) { c_kx = 0; ++c_ky; }
                conv_set_tap();
            }
        }
    };

    // epilogue vectors of this tile's BN columns, fetched now (one element per thread of group 0), parked in LDS after the loop
    float pre_bias = 0.f, pre_cs = 0.f;
    if constexpr (!TRANS) {
        if (kg == 0 && tid < BN && !tr) {
            if (p.bias) pre_bias = p.bias[n0 + tid];
            if (p.ln_stats) pre_cs = p.ln_colsum[n0 + tid];
        }
    }

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment reads: row = base16 + l15, logical chunk 4*kk + quad, physical chunk ^ ((row >> 1) & 7) (bases are multiples of 16)
    const int sw = (l15 >> 1) & 7;
    const int a_row_off = (wm * WTM + l15) * 128;
    const int b_row_off = A_BYTES + (wn * WTN + l15) * 128;

    const int nk = (p.K >> 6) / KS;   // K steps of this group (>= S - 1: dispatcher; eight-phase schedule: >= 2)

    // ---- eight-phase schedule (tile 42): LDS = two K-tile buffers of four HALF-TILES [A0 | A1 | WA | WB].  A-half h holds token
    // fragments [h MI/2, (h + 1) MI/2) of EVERY wave row (local row wm * PH_HR + i * 16 + l15; 128 rows = 16 KB), W part 0 / 1 the first NA /
    // last NB channel fragments of every wave column (local row wn * N? * 16 + j * 16 + l15): a half-tile is what ONE phase's MFMAs newly
    // need, so it is free again a phase after it was read and its successor (K-tile + 2, same buffer) can be on its way seven phases
    // before it is needed.  A K-tile is four phases (256 x 256: AI = 4, NA = NB = 2; 256 x 160: AI = 2, NA = 3, NB = 2):
    //     phase  fragment reads (ds_read_b128)      MFMAs                               half-tile staged
    //       0    A0 (2 AI) + WA (2 NA)               A0 x WA   (2 AI NA)                 WA of K-tile u + 1
    //       1    WB (2 NB)                           A0 x WB   (2 AI NB)                 A0 of K-tile u + 2
    //       2    A1 (2 AI)                           A1 x WB                             WB of K-tile u + 2
    //       3    WA (2 NA, again)                    A1 x WA                             A1 of K-tile u + 2   + the K-tile's ONE counted wait
    // Each phase is [reads + loads + lgkmcnt(0)] s_barrier [MFMAs] s_barrier; waves 4-7 run ONE barrier behind waves 0-3, so on every
    // SIMD (one wave of each half) one wave is in its MFMA segment while the other reads fragments and issues loads.  Hazards (interval =
    // the time between two consecutive barriers; half g's phase p reads in interval 2p + g):
    //   WAR  a half-tile's last reads are complete (lgkmcnt(0)) before the barrier that ends their interval, for both wave halves by the
    //        end of interval 2p + 1; its successor is issued in phase p + 1 or later (interval >= 2p + 2);
    //   RAW  every wave waits vmcnt(PH_INFLIGHT) in the read segment of phase 3 (after issuing that phase's loads): everything up to WA of
    //        the NEXT K-tile -- the youngest half-tile it needs -- has landed, three half-tiles stay in flight; waves 4-7 execute that wait
    //        one interval later, i.e. before the barrier that ends interval 8u + 7, and the first read of K-tile u + 1 is in interval 8u + 8.
    // Every wave issues the same number of loads per half-tile (a W part with fewer than 16 chunks: the second instruction of the waves
    // beyond it re-loads their first chunk), so one count is right for all of them.
    int ph8_wt0 = 1, ph8_wt1 = 0, ph8_ka = 0;    // K-tile indices of the next WA / WB half-tiles to stage; K offset (elements) of the next A half-tiles
    // K-tile index -> K offset of the weight rows.  Convolutions, korder = 1 (round 5): the K loop runs (64-channel chunk, tap) instead
    // of (tap, chunk) -- the nine taps of one channel chunk back to back.  A tile's nine taps read three input rows shifted by a pixel:
    // tap-major, 32 workgroups of an XCD stream 32 x 262 KB through a 4 MB L2 between two taps of the same row and every tap re-fetches
    // it (512 -> 512 @ 256^2: 607 MB fetched for 139 MB of operands, profiles/pmc_traffic.json); chunk-major, a workgroup's nine taps
    // touch 3 rows x 258 px x 128 B = 99 KB, 3.2 MB per XCD
    auto ph8_koff = [&](int t) {
        if constexpr (CONV) {
            if (p.korder) {
                const int ch = t / 9, tap = t - ch * 9;
                return tap * p.Cin + ch * 64;
            }
        }
        return t * 64;
    };
    // addresses as (wave-uniform 64-bit base) + (per-lane 32-bit byte offset, constant over the loop): the loads take the scalar-base form
    // and the loop keeps ONE offset register per operand instead of a strength-reduced 64-bit pointer per load (10 loads = 20 VGPRs on
    // the 256 x 320 tile, which spilled)
    const char* ph8_a = nullptr;
    const char* ph8_w = nullptr;
    unsigned ph8_a_lane = 0, ph8_w_lane = 0;
    if constexpr (PH8) {
        ph8_a = (const char*)(p.A + (size_t)(m0 + (PH_HR == 64 ? wave * 8 : (wave >> 2) * WTM + (wave & 3) * 8)) * p.lda);
        ph8_w = (const char*)(p.Wt + (size_t)n0 * p.K);
        ph8_a_lane = (unsigned)(lrow * p.lda + lchunk * 8) * 2u;
        ph8_w_lane = (unsigned)(lrow * p.K + lchunk * 8) * 2u;
    }
    auto ph8_stage_a = [&](int buf, int h) {
        char* dst = smem + buf * STAGE_BYTES + h * PH_A_HALF + wave * 1024;
#pragma unroll
        for (int qq = 0; qq < PH_A_Q; ++qq) {
            if constexpr (CONV) {
                const int off = ph8_toff[PH_A_Q * h + qq];
                const bf16_t* src = off >= 0 ? ph8_cbase + (off + c_cin0) : (const bf16_t*)g16_zero_page;
                glds16(src, dst + qq * 8192);
            } else {
                glds16(ph8_a + ((size_t)(qq * 128 + h * PH_HR) * p.lda + ph8_ka) * 2 + ph8_a_lane, dst + qq * 8192);
            }
        }
    };
    // Advance the A stream to the next K-tile.  Convolutions: BRANCH-FREE (scalar selects for the tap / chunk state in either K order,
    // `&` and selects for the halo test) -- it is issued inside an MFMA segment (below), and a branch would split that segment's basic
    // block and put all of this in front of the MFMAs instead of beside them
    auto ph8_adv_a = [&]() {
        if constexpr (CONV) {
            const int ko = p.korder;
            // chunk-major: next tap; after the ninth, the next 64-channel chunk.  tap-major: next chunk; after the last, the next tap
            const int a_kx1 = c_kx + 1, a_w1 = a_kx1 == 3, a_ky1 = c_ky + a_w1, a_w2 = a_ky1 == 3;
            const int b_c1 = c_cin0 + 64, b_w0 = b_c1 >= p.Cin, b_kx1 = c_kx + b_w0, b_w1 = b_kx1 == 3;
            const int n_kx = ko ? (a_w1 ? 0 : a_kx1) : (b_w1 ? 0 : b_kx1);
            const int n_ky = ko ? (a_w2 ? 0 : a_ky1) : c_ky + b_w1;
            const int n_cin = ko ? c_cin0 + (a_w2 ? 64 : 0) : (b_w0 ? b_c1 - p.Cin : b_c1);
            c_kx = n_kx;
            c_ky = n_ky;
            c_cin0 = n_cin;
            const int sh = p.up ? 1 : 0;
            const unsigned wl = (unsigned)(p.W * p.lda);      // < 2^24 (dispatcher): the 24-bit multiplier is full rate, the 32-bit one is not
#pragma unroll
            for (int q = 0; q < A_Q; ++q) {
                const int iy = c_iy0[q] + c_ky, ix = c_ix0[q] + c_kx;
                const int ok = ((unsigned)iy < (unsigned)VH) & ((unsigned)ix < (unsigned)VW);
                const unsigned off = __umul24((unsigned)(iy >> sh), wl) + __umul24((unsigned)(ix >> sh), (unsigned)p.lda);
                ph8_toff[q] = ok ? (int)off : -1;
            }
        } else {
            ph8_ka += 64;
        }
    };
    // W part `part` (0: the first NA fragments of every wave column, 1: the last NB): chunk c = wave + 8 qq holds local rows 8c .. 8c + 7 =
    // wave column c / (2 N?) , rows (c % (2 N?)) * 8 ...; a chunk index beyond the part re-loads chunk `wave`
    auto ph8_stage_w = [&](int buf, auto part_c, int koff) {
        constexpr int PART = decltype(part_c)::value;
        constexpr int NF = PART ? PH_NB : PH_NA, CH = PART ? PH_WB_CH : PH_WA_CH, NQ = PART ? PH_WB_Q : PH_WA_Q;
        char* dst = smem + buf * STAGE_BYTES + (PART ? PH_OFF_WB : PH_OFF_W);
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
            int c = wave + 8 * qq;
            c = c < CH ? c : wave;
            const int chan = (c / (2 * NF)) * WTN + (PART ? PH_NA * 16 : 0) + (c % (2 * NF)) * 8;
            glds16(ph8_w + ((size_t)chan * p.K + koff) * 2 + ph8_w_lane, dst + c * 1024);
        }
    };
    using PI0 = std::integral_constant<int, 0>;
    using PI1 = std::integral_constant<int, 1>;
    if constexpr (PH8) {
        // K-tile 0 complete, A0 / WB / A1 of K-tile 1 behind it (its WA goes out in phase 0 of K-tile 0)
        ph8_stage_a(0, 0);
        ph8_stage_w(0, PI0{}, 0);
        ph8_stage_w(0, PI1{}, 0);
        ph8_stage_a(0, 1);
        ph8_adv_a();
        ph8_stage_a(1, 0);
        ph8_stage_w(1, PI1{}, ph8_koff(1));
        ph8_stage_a(1, 1);
        ph8_adv_a();
        ph8_wt1 = 2;
    } else {
        if constexpr (HALO) {   // the halo tiles of chunks 0 and 1 first: the first counted wait leaves only the youngest W stage in flight
            halo_issue(0, 0, H_PQ);
            if (h_nchunks > 1) halo_issue(1, 0, H_PQ);
        }
#pragma unroll
        for (int s = 0; s < S - 1; ++s) {
#pragma unroll
            for (int q = 0; q < LOADS; ++q) stage_one(s, q);
            stage_advance();
        }
    }

    // LayerNorm folding: mean / rstd of the token rows this wave finishes after the K-group exchange (token fragments
    // i = kg*MIH + h).  Fetched and reduced HERE, under the first tiles' load latency, not in the epilogue (a chain of dependent
    // L2 / fabric round trips there).  The four lanes that share a token (one per quad) take a contiguous quarter of the
    // producer's slots each -- all loads issued before the first use -- and combine by two xor shuffles (fixed order).
    float ln_mean[MIH], ln_rstd[MIH];
#pragma unroll
    for (int h = 0; h < MIH; ++h) {
        ln_mean[h] = 0.f;
        ln_rstd[h] = 1.f;
    }
    if (p.ln_stats) {
        const int spq = (p.ln_slots + 3) >> 2;            // slots per quad (<= 8: dispatcher)
        const int s_lo = quad * spq;
#pragma unroll
        for (int h = 0; h < MIH; ++h) {
            const int m = m0 + wm * WTM + (kg * MIH + h) * 16 + l15;
            if (p.ln_slots == 0) {
                const float* st2 = p.ln_stats + (size_t)m * 2;
                ln_mean[h] = st2[0];
                ln_rstd[h] = st2[1];
                continue;
            }
            const supir_f32x2* st = (const supir_f32x2*)(p.ln_stats + (size_t)m * p.ln_ld * 2);
            supir_f32x2 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {                   // clamped index + zero weight instead of a branch around the load
                const int sl = s_lo + (e < spq ? e : spq - 1);   // beyond this quad's share: repeat its last slot (cache hit)
                v[e] = st[sl < p.ln_slots ? sl : p.ln_slots - 1];
            }
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = e < spq && s_lo + e < p.ln_slots;
                sm += ok ? v[e][0] : 0.f;
                sq += ok ? v[e][1] : 0.f;
            }
            sm += __shfl_xor(sm, 16, 64);
            sq += __shfl_xor(sq, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float inv = 1.0f / (float)p.K;
            const float mean = sm * inv;
            float var = sq * inv - mean * mean;
            var = var > 0.f ? var : 0.f;
            ln_mean[h] = mean;
            ln_rstd[h] = rsqrtf(var + p.ln_eps);
        }
    }

    // Residual / row-bias operands of the epilogue (this lane's 4 channels of its token per fragment), fetched NOW: issued in the
    // epilogue they were a dependent L2 / fabric round trip (~0.5-1 us) at the very end of a 13 us launch.  In-place residuals
    // (x += f(x)) are safe: a workgroup reads its whole tile here, long before any of its stores, and tiles are disjoint.
    // (only where it is cheap in registers: 2 x MIH x NI x 2 VGPRs held across the main loop; the 256 x 160 tile would spill)
    constexpr bool PRE = !TRANS && MIH * NI <= 10;
    u32x2 pre_res[PRE ? MIH : 1][PRE ? NI : 1], pre_rb[PRE ? MIH : 1][PRE ? NI : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int h = 0; h < MIH; ++h)
#pragma unroll
            for (int j = 0; j < NI; ++j) pre_res[h][j] = pre_rb[h][j] = u32x2{0u, 0u};
        if (!tr && p.act != 2) {
#pragma unroll
            for (int h = 0; h < MIH; ++h) {
                const int m = m0 + wm * WTM + (kg * MIH + h) * 16 + l15;
                if (p.res) {
                    const bf16_t* rp = p.res + (size_t)m * p.ldr + n0 + wn * WTN + 4 * quad;
#pragma unroll
                    for (int j = 0; j < NI; ++j) pre_res[h][j] = *(const u32x2*)(rp + j * 16);
                }
                if (p.rowbias) {
                    const bf16_t* rbp = p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rb + n0 + wn * WTN + 4 * quad;
#pragma unroll
                    for (int j = 0; j < NI; ++j) pre_rb[h][j] = *(const u32x2*)(rbp + j * 16);
                }
            }
        }
    }

    // one K step.  STAGE: also issue the global->LDS loads of tile kt + S - 1 (into the buffer step kt - 1 just finished reading),
    // spread over the two 32-wide K slices; INFLIGHT: younger stages that may stay outstanding across this step's barrier
    int buf = 0, sbuf = S - 1;   // ring positions of the tile being read / being staged (kept modulo S without a division)
#ifdef SUPIR_G16_TIMELINE
    unsigned long long tl_s0 = 0, tl_s1 = 0, tl_s2 = 0, tl_s3 = 0, tl_wait = 0, tl_bar = 0, tl_comp = 0;
#endif
    auto kstep = [&](auto stage_c, auto inflight_c, auto trans_c) {
        constexpr bool STAGE = decltype(stage_c)::value;
        constexpr int INFLIGHT = decltype(inflight_c)::value;
        constexpr bool TR = decltype(trans_c)::value;
        G16_TLS(tl_s0);
        if constexpr (HALO && INFLIGHT == 1) {   // the previous iteration's loads stay in flight: its W chunks + the halo pieces it issued
            if (h_cnt_prev == 0) g16_wait_vmcnt<LOADS>();
            else if (h_cnt_prev == 1) g16_wait_vmcnt<LOADS + 1>();
            else g16_wait_vmcnt<LOADS + 2>();
        } else {
            static_assert(!HALO || INFLIGHT == 0, "halo form: rings of depth 2 or 3");
            g16_wait_vmcnt<INFLIGHT * LOADS>();
        }
        G16_TLS(tl_s1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        G16_TLS(tl_s2);
        const char* sT = ring + buf * STAGE_BYTES;
        // halo form: this step's tap shift inside the halo tile of its chunk
        const int h_ky = (h_rtap * 11) >> 5, h_toff = h_ky * H_RS + (h_rtap - 3 * h_ky);
        const char* h_buf = smem + H_OFF + (h_rchunk & 1) * H_BYTES;
        // both 32-wide K slices of fragments are read ahead of the MFMAs -- except where the accumulators already take half the
        // register file (256 x 256: MI * NI = 32 fragments = 128 registers; 96 more for two slices spilled): one slice at a time
        // there, the second one read behind the first slice's MFMAs (the SIMD's other wave covers the wait)
        constexpr bool ONE_SLICE = MI * NI >= 32;
        constexpr int FS = ONE_SLICE ? 1 : 2;
        bf16x8 af[FS][MI], bfr[FS][NI];
        auto read_frags = [&](int kk, int slot) {
            const int coff = ((4 * kk + quad) ^ sw) * 16;
            if constexpr (HALO) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int P = h_p0[i] + h_toff;
                    af[slot][i] = *(const bf16x8*)(h_buf + P * 128 + ((((4 * kk + quad) ^ P) & 7) << 4));
                }
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[slot][i] = *(const bf16x8*)(sT + a_row_off + i * 16 * 128 + coff);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[slot][j] = *(const bf16x8*)(sT + b_row_off + j * 16 * 128 + coff);
        };
        read_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sl = ONE_SLICE ? 0 : kk;
            if (!ONE_SLICE && kk == 0) read_frags(1, 1);
            if constexpr (STAGE) {
#pragma unroll
                for (int q = (kk * LOADS) / 2; q < ((kk + 1) * LOADS) / 2; ++q) stage_one(sbuf, q);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if constexpr (TR)
                        acc[i][j] = SUPIR_MFMA_16x16x32(af[sl][i], bfr[sl][j], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = SUPIR_MFMA_16x16x32(bfr[sl][j], af[sl][i], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (ONE_SLICE && kk == 0) read_frags(1, 0);
        }
        if constexpr (STAGE) stage_advance();
        if constexpr (HALO) {
            int cnt = 0;
            if constexpr (STAGE) {
                if (h_next < h_nchunks && h_it >= h_it0) {     // wave-uniform
                    const int j = h_it - h_it0;
#pragma unroll
                    for (int J = 0; J < H_NLI; ++J)
                        if (j == J) cnt = halo_issue(h_next, J * H_HQ, (J + 1) * H_HQ);
                    if (j == H_NLI - 1) {
                        ++h_next;
                        h_it0 = (9 * (h_next - 2) + 8) / KS + 1;
                    }
                }
            }
            h_cnt_prev = cnt;
            ++h_it;
            h_rtap += KS;
            if (h_rtap >= 9) { h_rtap -= 9; ++h_rchunk; }
        }
        buf = buf + 1 == S ? 0 : buf + 1;
        sbuf = sbuf + 1 == S ? 0 : sbuf + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G16_TLS(tl_s3);
        G16_TLACC(tl_wait, tl_s0, tl_s1);
        G16_TLACC(tl_bar, tl_s1, tl_s2);
        G16_TLACC(tl_comp, tl_s2, tl_s3);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    G16_TL(tl_loop0);
    auto main_loop = [&](auto trans_c) {
        for (int kt = 0; kt + S - 1 < nk; ++kt) kstep(T_{}, std::integral_constant<int, S - 2>{}, trans_c);
        if constexpr (S == 3) kstep(F_{}, std::integral_constant<int, 1>{}, trans_c);
        kstep(F_{}, std::integral_constant<int, 0>{}, trans_c);
    };
    if constexpr (PH8) {
        const int fa_off = (wm * PH_HR + l15) * 128;
        const int fw_off0 = PH_OFF_W + (wn * PH_NA * 16 + l15) * 128, fw_off1 = PH_OFF_WB + (wn * PH_NB * 16 + l15) * 128;
        const int grp = bwave >> 2;     // waves 4-7 run one barrier behind waves 0-3 (wave w and w + 4 share a SIMD)
        int cur = 0;
        // one K-tile.  SW0: stage WA of the next K-tile (phase 0); SA: stage A0 / WB / A1 of the K-tile after it (phases 1 - 3);
        // WAITN: the counted wait of phase 3 (PH_INFLIGHT = three half-tiles stay in flight; 0 = the tail; -1 = nothing left to wait for);
        // TR: the transposed epilogue's operand order (fused q|k|v: the V^T column tiles)
        auto ph8_tile = [&](auto sw0_c, auto sa_c, auto wait_c, auto tr_c) {
            constexpr bool SW0 = decltype(sw0_c)::value, SA = decltype(sa_c)::value, TR = decltype(tr_c)::value;
            constexpr int WAITN = decltype(wait_c)::value;
            const char* sT = smem + cur * STAGE_BYTES;
            bf16x8 af[2][PH_AI], wf[2][PH_NA];
            auto read_a = [&](int h) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < PH_AI; ++i) af[kk][i] = *(const bf16x8*)(sT + h * PH_A_HALF + fa_off + i * 2048 + (((4 * kk + quad) ^ sw) * 16));
            };
            auto read_w = [&](auto part_c) {
                constexpr int PART = decltype(part_c)::value;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < (PART ? PH_NB : PH_NA); ++j)
                        wf[kk][j] = *(const bf16x8*)(sT + (PART ? fw_off1 : fw_off0) + j * 2048 + (((4 * kk + quad) ^ sw) * 16));
            };
            auto mid = [&]() {     // end of a read segment: this wave's fragment reads are complete before anyone passes the barrier
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            // shadow_c: the A stream's advance to the next K-tile (convolutions: the tap addresses of this lane's four rows, ~40 VALU
            // instructions -- every K-tile in the chunk-major order) issued INSIDE this MFMA segment, three VALU per MFMA: the matrix pipe
            // takes an instruction every ~16 cycles and the vector pipe is free beside it; in the read segment the same work held up
            // the barrier the partner wave's MFMAs wait behind (first version: -16 %, profiles/r05/experiment_tile42_conv_chunk_major_*)
            auto mma = [&](auto ah_c, auto part_c, auto shadow_c) {
                constexpr int AH = decltype(ah_c)::value, PART = decltype(part_c)::value;
                constexpr bool SHADOW = decltype(shadow_c)::value;
                __builtin_amdgcn_s_setprio(1);
                if constexpr (SHADOW) ph8_adv_a();
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < PH_AI; ++i)
#pragma unroll
                        for (int j = 0; j < (PART ? PH_NB : PH_NA); ++j) {
                            f32x4_t& c = acc[AH * PH_AI + i][(PART ? PH_NA : 0) + j];
                            if constexpr (TR) c = SUPIR_MFMA_16x16x32(af[kk][i], wf[kk][j], c, 0, 0, 0);
                            else c = SUPIR_MFMA_16x16x32(wf[kk][j], af[kk][i], c, 0, 0, 0);
                        }
                if constexpr (SHADOW && CONV) {
#pragma unroll
                    for (int g = 0; g < 2 * PH_AI * (PART ? PH_NB : PH_NA); ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);    // up to two SALU (the tap / chunk state) ...
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);    // ... and three VALU (the four rows' tap offsets) behind it
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            // phase 0
            read_w(PI0{});
            read_a(0);
            if constexpr (SW0) ph8_stage_w(cur ^ 1, PI0{}, ph8_koff(ph8_wt0));
            mid();
            mma(PI0{}, PI0{}, F_{});
            // phase 1
            read_w(PI1{});
            if constexpr (SA) ph8_stage_a(cur, 0);
            mid();
            mma(PI0{}, PI1{}, F_{});
            // phase 2
            read_a(1);
            if constexpr (SA) ph8_stage_w(cur, PI1{}, ph8_koff(ph8_wt1));
            mid();
            mma(PI1{}, PI1{}, F_{});
            // phase 3
            read_w(PI0{});
            if constexpr (SA) ph8_stage_a(cur, 1);
            if constexpr (WAITN >= 0) g16_wait_vmcnt<WAITN>();
            mid();
            mma(PI1{}, PI0{}, std::integral_constant<bool, SA>{});
            ++ph8_wt0;
            ++ph8_wt1;
            cur ^= 1;
        };
        g16_wait_vmcnt<PH_INFLIGHT>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (grp == 1) {      // the second wave half runs one barrier behind the first from here on
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        using WF_ = std::integral_constant<int, PH_INFLIGHT>;
        using W0_ = std::integral_constant<int, 0>;
        using WN_ = std::integral_constant<int, -1>;
        auto ph8_loop = [&](auto tr_c) {
            for (int u = 0; u + 2 < nk; ++u) ph8_tile(T_{}, T_{}, WF_{}, tr_c);
            ph8_tile(T_{}, F_{}, W0_{}, tr_c);
            ph8_tile(F_{}, F_{}, WN_{}, tr_c);
        };
        if constexpr (MIXED) {
            if (tr) ph8_loop(T_{});
            else ph8_loop(F_{});
        } else {
            ph8_loop(F_{});
        }
        if (grp == 0) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    } else if constexpr (MIXED) {
        if (tr) main_loop(T_{});
        else main_loop(F_{});
    } else {
        main_loop(std::integral_constant<bool, TRANS>{});
    }

    // ------------------------------------------------------------------ epilogue
    G16_TL(tl_loop1);
    // (Round 5 tried fetching the residual / row-bias operands of the large tiles AHEAD of their 16-token block -- all blocks at once on
    // 256 x 160, two in flight on 256 x 256: the per-wave epilogue fell from 14.3 k to 6.4 k cycles and the wait moved in front of the
    // barrier, 0.5 k -> 11 k cycles; launches +2..5 % slower.  256 workgroups pull 80 KB of residual and push 80 KB of output each in the
    // same few microseconds: the epilogue of the large tiles is THROUGHPUT-bound, not a chain of round trips.
    // profiles/r05/experiment_timeline_epilogue_prefetch_and_tile43.log)
    __syncthreads();   // every wave is done with its last fragment reads: the rings are scratch from here on
    if constexpr (!TRANS) {
        if (kg == 0 && tid < BN && !tr) {
            ((float*)(smem + OFF_BIAS))[tid] = pre_bias;
            ((float*)(smem + OFF_BIAS))[BN + tid] = pre_cs;
        }
    }
    if constexpr (KS == 2) {
        // reduce-scatter of the K partials: group g keeps token fragments [g*MIH, (g+1)*MIH) and adds the other group's partial.
        // layout [sender][wave][h][j][r][lane] fp32: lane-contiguous, conflict-free
        float* xch = (float*)smem;
        const int w_off = wave * (MIH * NI * 4 * 64) + lane;
        float* mine = xch + kg * (XCH_HALF / 4) + w_off;
        const float* theirs = xch + (1 - kg) * (XCH_HALF / 4) + w_off;
        if (kg == 0) {
#pragma unroll
            for (int h = 0; h < MIH; ++h)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((h * NI + j) * 4 + r) * 64] = acc[MIH + h][j][r];
        } else {
#pragma unroll
            for (int h = 0; h < MIH; ++h)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((h * NI + j) * 4 + r) * 64] = acc[h][j][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int h = 0; h < MIH; ++h)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[h][j][r] += theirs[((h * NI + j) * 4 + r) * 64];
        } else {
#pragma unroll
            for (int h = 0; h < MIH; ++h)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[MIH + h][j][r] += theirs[((h * NI + j) * 4 + r) * 64];
        }
    } else {
        __syncthreads();   // bias / column sums are in LDS
    }
    G16_TL(tl_xch);
    // from here on the wave's data is acc[kg*MIH + h][j], h < MIH (compile-time indices in both wave-uniform branches)
    auto prefetch_next = [&]() {
        const unsigned pf_lines = p.pf_lines;
        if (pf_lines == 0) return;
        const unsigned total_waves = gridDim.x / NP * NWT, gw = (unsigned)(vidx * NX + vxcd) * NWT + bwave;   // per problem
        const unsigned n_instr = (pf_lines + 63) >> 6;
        for (unsigned i = gw; i < n_instr; i += total_waves) {
            unsigned line = i * 64 + lane;
            line = line < pf_lines ? line : pf_lines - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.pf_ptr + (size_t)line * 128),
                                             (__attribute__((address_space(3))) void*)(smem + (PH8 ? RING - 256 : KS * RING + (HALO ? 2 * H_BYTES : 0))), 4, 0, 0);
        }
    };
    using K0_ = std::integral_constant<int, 0>;
    using K1_ = std::integral_constant<int, 1>;

    if constexpr (TRANS || MIXED) {
      if (tr) {
        // D[i = token][j = channel]: lane owns channel l15 of fragment j, tokens 4*quad + r of fragment i -> 4 consecutive tokens
        bf16_t* Cb = (bf16_t*)(MIXED ? p.C2 : p.C);
        const int ldc_t = MIXED ? p.ldc2 : p.ldc, n_base = MIXED ? p.n_split : 0, n_out_total = p.N - n_base;
        auto run = [&](auto kg_c) {
            constexpr int KG = decltype(kg_c)::value;
#pragma unroll
            for (int h = 0; h < MIH; ++h) {
                const int mb = m0 + wm * WTM + (KG * MIH + h) * 16 + 4 * quad;      // first of this lane's 4 tokens
                const int b = mb / p.rows_per_batch, t = mb - b * p.rows_per_batch;   // rows_per_batch % 4 == 0 (dispatcher)
                float mu[4], rs[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // statistics of token 4*quad + r live in the lanes with that l15
                    mu[r] = __shfl(ln_mean[h], 4 * quad + r, 64);
                    rs[r] = __shfl(ln_rstd[h], 4 * quad + r, 64);
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int n = n0 + wn * WTN + j * 16 + l15;
                    const float bz = p.bias ? p.bias[n] : 0.f;
                    const float cs = p.ln_stats ? p.ln_colsum[n] : 0.f;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = p.alpha * (rs[r] * (acc[KG * MIH + h][j][r] - mu[r] * cs) + bz);
                    const u32x2 o = {f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3])};
                    supir_store8(Cb + ((size_t)b * n_out_total + (n - n_base)) * ldc_t + t, o);
                }
            }
        };
        if (kg == 0) run(K0_{});
        else if constexpr (KS == 2) run(K1_{});
        prefetch_next();
        return;
      }
    }
    if constexpr (!TRANS) {
        const float* s_bias = (const float*)(smem + OFF_BIAS);
        const float* s_cs = s_bias + BN;
        char* c_stage = smem + OFF_CST + bwave * C_STAGE;
        // 16-byte pieces of a staged [16][COLS] bf16 block -> row-contiguous global stores
        auto flush_block = [&](int row_base, int col_base, auto cols_c) {
            constexpr int COLS = decltype(cols_c)::value;
            constexpr int PPR = COLS / 8, PIECES = 16 * PPR;
#pragma unroll
            for (int rr = 0; rr < (PIECES + 63) / 64; ++rr) {
                const int id = rr * 64 + lane;
                if (PIECES % 64 == 0 || id < PIECES) {
                    const int row = id / PPR, ch = id - row * PPR;
                    const f32x4 piece = *(const f32x4*)(c_stage + row * C_RS + ch * 16);
                    supir_store16((bf16_t*)p.C + (size_t)(row_base + row) * p.ldc + col_base + ch * 8, piece);
                }
            }
        };
        if (p.act == 2) {
            // GEGLU: W rows interleaved [16 value | 16 gate]: fragment pair (2 jp, 2 jp + 1) = (value, gate) of 16 output columns
            if constexpr ((NI & 1) == 0) {
                auto run = [&](auto kg_c) {
                    constexpr int KG = decltype(kg_c)::value;
#pragma unroll
                    for (int h = 0; h < MIH; ++h) {
                        const int mrow = wm * WTM + (KG * MIH + h) * 16;
                        const float mu = ln_mean[h], rs = ln_rstd[h];
#pragma unroll
                        for (int jp = 0; jp < NI / 2; ++jp) {
                            const int nl = wn * WTN + jp * 32 + 4 * quad;
                            const f32x4 bv = *(const f32x4*)(s_bias + nl), bg = *(const f32x4*)(s_bias + nl + 16);
                            const f32x4 cv = *(const f32x4*)(s_cs + nl), cg = *(const f32x4*)(s_cs + nl + 16);
                            float r[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = rs * (acc[KG * MIH + h][2 * jp][e] - mu * cv[e]) + bv[e];
                                const float g = rs * (acc[KG * MIH + h][2 * jp + 1][e] - mu * cg[e]) + bg[e];
                                r[e] = v * (p.fast_gelu ? gelu_fast_f(g) : gelu_f(g));
                            }
                            const u32x2 o = {f2bf_pk(r[0], r[1]), f2bf_pk(r[2], r[3])};
                            *(u32x2*)(c_stage + l15 * C_RS + (jp * 16 + 4 * quad) * 2) = o;
                        }
                        flush_block(m0 + mrow, (n0 >> 1) + wn * (WTN / 2), std::integral_constant<int, WTN / 2>{});
                    }
                };
                if (kg == 0) run(K0_{});
                else if constexpr (KS == 2) run(K1_{});
            }
            prefetch_next();
            return;
        }
        auto run = [&](auto kg_c, auto silu_c) {
            constexpr int KG = decltype(kg_c)::value;
            constexpr bool SILU = decltype(silu_c)::value;
            constexpr int GP = (WTN + 63) / 64;
            float gsum[GP], gsq[GP];   // GroupNorm statistics from the producer: this lane's columns, summed over the wave's token blocks
#pragma unroll
            for (int g = 0; g < GP; ++g) gsum[g] = gsq[g] = 0.f;
#pragma unroll
            for (int h = 0; h < MIH; ++h) {
                const int mrow = wm * WTM + (KG * MIH + h) * 16;    // first token of this 16-token block inside the tile
                const int m = m0 + mrow + l15;
                const float mu = ln_mean[h], rs = ln_rstd[h];
                u32x2 e_rb[NI], e_res[NI];
                if constexpr (PRE) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        e_rb[j] = pre_rb[h][j];
                        e_res[j] = pre_res[h][j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NI; ++j) e_rb[j] = e_res[j] = u32x2{0u, 0u};
                    if (p.rowbias) {
                        const bf16_t* rbp = p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rb + n0 + wn * WTN + 4 * quad;
#pragma unroll
                        for (int j = 0; j < NI; ++j) e_rb[j] = *(const u32x2*)(rbp + j * 16);
                    }
                    if (p.res) {
                        const bf16_t* rp = p.res + (size_t)m * p.ldr + n0 + wn * WTN + 4 * quad;
#pragma unroll
                        for (int j = 0; j < NI; ++j) e_res[j] = *(const u32x2*)(rp + j * 16);
                    }
                }
                float rsum = 0.f, rsq = 0.f;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int nl = wn * WTN + j * 16 + 4 * quad;
                    const f32x4 cs = *(const f32x4*)(s_cs + nl), bz = *(const f32x4*)(s_bias + nl);
                    const u32x2 rb = e_rb[j], rr = e_res[j];
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = rs * (acc[KG * MIH + h][j][e] - mu * cs[e]) + bz[e];
                        t += (e & 1) ? bfhi2f(rb[e >> 1]) : bflo2f(rb[e >> 1]);
                        if constexpr (SILU) t = p.act == 1 ? silu_f(t) : (p.act == 3 ? gelu_f(t) : quick_gelu_f(t));
                        t *= p.alpha;
                        t += (e & 1) ? bfhi2f(rr[e >> 1]) : bflo2f(rr[e >> 1]);
                        v[e] = t;
                    }
                    const u32x2 o = {f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3])};
                    const float r0 = bflo2f(o[0]), r1 = bfhi2f(o[0]), r2 = bflo2f(o[1]), r3 = bfhi2f(o[1]);
                    rsum += (r0 + r1) + (r2 + r3);
                    rsq += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
                    *(u32x2*)(c_stage + l15 * C_RS + (j * 16 + 4 * quad) * 2) = o;
                }
                flush_block(m0 + mrow, n0 + wn * WTN, std::integral_constant<int, WTN>{});
                if (p.gn_part_out) {   // column sums of the staged 16 x WTN block, of the bf16 values as stored: lane = column
#pragma unroll
                    for (int g = 0; g < GP; ++g) {
                        const int col = g * 64 + lane;
                        if (WTN % 64 == 0 || col < WTN) {
                            float cs_ = 0.f, cq_ = 0.f;
#pragma unroll
                            for (int row = 0; row < 16; ++row) {
                                const float f = bf2f(*(const u16*)(c_stage + row * C_RS + col * 2));
                                cs_ += f;
                                cq_ += f * f;
                            }
                            gsum[g] += cs_;
                            gsq[g] += cq_;
                        }
                    }
                }
                if (p.rowstats_out) {
                    rsum += __shfl_xor(rsum, 16, 64);
                    rsq += __shfl_xor(rsq, 16, 64);
                    rsum += __shfl_xor(rsum, 32, 64);
                    rsq += __shfl_xor(rsq, 32, 64);
                    if (quad == 0) {
                        if constexpr (WN == 1) {   // the wave covers the tile's whole column range: this IS the tile's slot
                            float* dst = p.rowstats_out + ((size_t)m * p.rs_ld + tile_n) * 2;
                            dst[0] = rsum;
                            dst[1] = rsq;
                        } else {
                            float* red = (float*)(smem + OFF_RED) + ((size_t)wn * BM + mrow + l15) * 2;
                            red[0] = rsum;
                            red[1] = rsq;
                        }
                    }
                }
            }
            if (p.gn_part_out) {
                float* red = (float*)(smem + OFF_GN) + (size_t)bwave * WTN * 2;
#pragma unroll
                for (int g = 0; g < GP; ++g) {
                    const int col = g * 64 + lane;
                    if (WTN % 64 == 0 || col < WTN) {
                        red[col * 2] = gsum[g];
                        red[col * 2 + 1] = gsq[g];
                    }
                }
            }
        };
        if (kg == 0) {
            if (p.act != 0) run(K0_{}, T_{});
            else run(K0_{}, F_{});
        } else if constexpr (KS == 2) {
            if (p.act != 0) run(K1_{}, T_{});
            else run(K1_{}, F_{});
        }
        if constexpr (WN > 1) {
            if (p.rowstats_out) {   // combine the wave columns in a fixed order: one slot per tile column
                __syncthreads();
                for (int r = (int)threadIdx.x; r < BM; r += NTHREADS) {
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) {
                        sm += ((const float*)(smem + OFF_RED))[((size_t)w * BM + r) * 2];
                        sq += ((const float*)(smem + OFF_RED))[((size_t)w * BM + r) * 2 + 1];
                    }
                    float* dst = p.rowstats_out + ((size_t)(m0 + r) * p.rs_ld + tile_n) * 2;
                    dst[0] = sm;
                    dst[1] = sq;
                }
            }
        }
        if (p.gn_part_out) {
            // one (sum, sum of squares) per GU-channel unit of this tile: every wave that covered the unit's columns (all wave rows,
            // both K groups), in a fixed order (reproducible); tile rows never straddle a batch (dispatcher).  GU = 10 for the
            // 80 / 160-column tiles (the UNet's groups are 10 / 20 / 40 / 60 channels wide), 4 for the 128 / 256-column tiles 39 / 40
            // (the VAE's groups: 4 / 8 / 16 channels, sgm/modules/diffusionmodules/model.py:48-51)
            constexpr int GU = (BN % 10 == 0) ? 10 : 4;
            __syncthreads();
            if ((int)threadIdx.x < BN / GU) {
                const float* red = (const float*)(smem + OFF_GN);
                float sm = 0.f, sq = 0.f;
                for (int c = (int)threadIdx.x * GU; c < (int)threadIdx.x * GU + GU; ++c) {
                    const int wn_c = c / WTN, cl = c - wn_c * WTN;
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                        for (int wx = 0; wx < WM; ++wx) {
                            const int bw = kx * NW + wx * WN + wn_c;
                            sm += red[((size_t)bw * WTN + cl) * 2];
                            sq += red[((size_t)bw * WTN + cl) * 2 + 1];
                        }
                }
                const int b = m0 / p.rows_per_batch, chunk = (m0 - b * p.rows_per_batch) / BM, nchunk = p.rows_per_batch / BM;
                float* dst = p.gn_part_out + (((size_t)b * nchunk + chunk) * (p.N / GU) + n0 / GU + threadIdx.x) * 2;
                dst[0] = sm;
                dst[1] = sq;
            }
        }
#ifdef SUPIR_G16_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (g16_tl_buf && lane == 0) {
            const unsigned long long tl_end = __builtin_amdgcn_s_memtime();
            unsigned long long* o = g16_tl_buf + ((size_t)blockIdx.x * 8 + bwave) * 16;
            o[8] = tl_wait;
            o[9] = tl_bar;
            o[10] = tl_comp;
            o[0] = tl_start;
            o[1] = tl_loop0 - tl_start;
            o[2] = tl_loop1 - tl_loop0;
            o[3] = tl_xch - tl_loop1;
            o[4] = tl_end - tl_xch;
            o[5] = tl_end - tl_start;
            o[6] = tl_end;
        }
#endif
        prefetch_next();
    }
}

template <int BM, int BN, int WM, int WN, int KS, int S, bool TRANS, bool CONV = false, bool MIXED = false, int NP = 1, int HW_ = 0>
static int launch_gemm16(const GemmArgs* a_in, hipStream_t st) {
    GemmArgsN<NP> pp;
    for (int q = 0; q < NP; ++q) {
        pp.p[q] = a_in[q];
        pp.p[q].korder = (CONV && S == 8 && supir_debug_knob_value(6) != 1) ? 1 : 0;   // tile 42 / 45 convolutions: chunk-major K order (knob 6 = 1: tap-major)
    }
    GemmArgs& a = pp.p[0];
    const int tiles = (a.M / BM) * (a.N / BN);
    if (NP > 1 && tiles % (8 / NP)) return SUPIR_ERR_SHAPE;
    const double a_bytes = CONV ? 2.0 * (double)a.M * (a.up ? 0.25 : (double)(a.stride * a.stride)) * a.Cin : 2.0 * (double)a.M * a.K;
    supir_choose_xcd_grid(a, a.M / BM, a.N / BN, a_bytes, 2.0 * (double)a.N * a.K, CONV ? (HW_ > 0 ? 2 : 9) : 1, 1, 8 / NP);
    for (int q = 1; q < NP; ++q) {   // identical shapes: identical tile maps
        pp.p[q].gm = a.gm;
        pp.p[q].gn = a.gn;
        pp.p[q].order = a.order;
    }
    // the ring(s) + the prefetch scratch row (the eight-phase tiles dump the prefetch into the last 256 bytes of their ring, which the
    // epilogue scratch never reaches: the 512 x 128 tile uses all 160 KB)
    // halo form: the rings hold W tiles only, two halo tiles of ((BM / W + 2) (W + 2) pixels, rounded up to 8) x 128 B behind them
    constexpr int halo_bytes = HW_ > 0 ? 2 * ((((BM / (HW_ > 0 ? HW_ : 1) + 2) * (HW_ + 2) + 7) / 8) * 1024) : 0;
    constexpr int smem = HW_ > 0 ? KS * S * BN * 128 + halo_bytes + 256 : KS * (S == 8 ? 2 : S) * (BM + BN) * 128 + (S == 8 ? 0 : 256);
    static_assert(smem <= 163840, "LDS");
    auto kern = gemm16_kernel<BM, BN, WM, WN, KS, S, TRANS, CONV, MIXED, NP, HW_>;
    static bool attr_set = false;
    if (!attr_set) {
        if (supir_note_hip_status(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) != SUPIR_OK) return SUPIR_ERR_HIP;
        attr_set = true;
    }
    SUPIR_LAUNCH(kern, dim3(NP * tiles), dim3(64 * WM * WN * KS), smem, st, pp);
    return SUPIR_LAUNCH_STATUS();
}

// tiles 32 (128 x 80), 33 (128 x 160), 34 (256 x 160, GEGLU-capable), 35 (128 x 80, 3-deep rings): exact fits only
bool supir_gemm16_supported(const GemmArgs& a, int tile, bool conv) {
    if ((tile < 32 || tile > 35) && (tile < 38 || tile > 40) && tile != 42 && tile != 45 && (tile < 48 || tile > 51)) return false;
    const bool halo = tile >= 48;       // LDS-staged halo convolutions: 48 = 128 x 80 @ W 32, 49 = 128 x 160 @ W 64, 50 = 256 x 160 @ W 32, 51 = 256 x 160 @ W 64
    const bool wide = tile == 34 || tile == 39 || tile == 40 || tile == 42 || tile == 45 || tile == 50 || tile == 51;
    const int bm = tile == 45 ? 512 : wide ? 256 : 128;
    const int bn = (tile == 39 || tile == 45) ? 128 : (tile == 40 || tile == 42) ? 256 : (tile == 32 || tile == 35 || tile == 38 || tile == 48) ? 80 : 160;
    const int ks = (wide || tile == 38) ? 1 : 2, s = (tile == 32 || tile == 33 || tile == 40 || tile == 49 || tile == 51) ? 2 : 3;   // tiles 42 / 45: at least two K-tiles
    if (halo) {
        const int hw = (tile == 49 || tile == 51) ? 64 : 32;
        // whole map rows per tile, stride 1, pad 1, no upsampling; 32-bit element offsets inside one image
        if (!conv || a.W != hw || a.OW != hw || a.OH != a.H || a.stride != 1 || a.up || a.pad_t != 1 || a.pad_l != 1) return false;
        if (a.rows_per_batch != a.OH * a.OW || a.rows_per_batch % bm || (long)a.H * a.W * a.lda >= (1L << 31)) return false;
    }
    if (a.M % bm || a.N % bn || a.K % (64 * ks) || a.lda % 8 || (a.K >> 6) / ks < s - 1) return false;
    if (a.out_mode == 1 || a.ln_slots > 32) return false;
    // tiles 39 / 40: plain and convolution forms with the ordinary epilogue only (no transposed output, no GEGLU); their GroupNorm
    // partials come in 4-channel units (channel counts 128 / 256 / 512: 4 / 8 / 16 channels per group)
    if ((tile == 39 || tile == 40 || tile == 42 || tile == 45) && (a.out_mode != 0 || a.act == 2)) return false;
    if (a.gn_part_out && (a.out_mode != 0 || a.act == 2 || a.rows_per_batch <= 0 || a.rows_per_batch % bm || a.N % (bn % 10 == 0 ? 10 : 4))) return false;
    if (conv) {
        // tile 42 keeps 32-bit tap offsets against ONE image base per tile: rows of a tile in one batch element, the image below 2^31 elements
        if ((tile == 42 || tile == 45) && (a.rows_per_batch % bm || (long)a.H * a.W * a.lda >= (1L << 31) || (long)a.W * a.lda >= (1L << 24) || a.H >= (1 << 23))) return false;
        if (a.Cin % (64 * ks) || a.K != 9 * a.Cin || a.out_mode != 0 || a.act == 2 || a.ln_stats || a.rowstats_out) return false;
    }
    if (a.act == 2)
        return tile == 34 && a.out_mode == 0 && !a.res && !a.rowbias && !a.rowstats_out && a.ldc % 8 == 0 && (((size_t)a.C) & 15) == 0;
    if (a.out_mode == 2) return a.rows_per_batch % 4 == 0 && a.ldc % 4 == 0 && !a.res && !a.rowbias && a.act == 0;
    if (a.ldc % 8 || (((size_t)a.C) & 15)) return false;
    if ((a.res && a.ldr % 4) || (a.rowbias && a.ld_rb % 4)) return false;
    return true;
}

int supir_gemm16_launch(const GemmArgs& a, hipStream_t st, int tile, bool conv) {
    if (!supir_gemm16_supported(a, tile, conv)) return SUPIR_ERR_SHAPE;
    if (conv) {
        switch (tile) {
            case 32: return launch_gemm16<128, 80, 4, 1, 2, 2, false, true>(&a, st);
            case 33: return launch_gemm16<128, 160, 2, 2, 2, 2, false, true>(&a, st);
            // 256 x 160: the eight waves as 4 x 2 (64 x 80 per wave: 18 fragment reads per K step instead of 24 for 8 x 1).  Same tile,
            // same K order: outputs bitwise equal; +3..5 % on the large convolutions (profiles/r04/micro_256x160_waves_*.log).
            // knob 1 = 2 forces 8 x 1 (A/B runs)
            case 34: return supir_debug_knob_value(1) != 2 ? launch_gemm16<256, 160, 4, 2, 1, 3, false, true>(&a, st)
                                                           : launch_gemm16<256, 160, 8, 1, 1, 3, false, true>(&a, st);
            case 38: return launch_gemm16<128, 80, 4, 1, 1, 3, false, true>(&a, st);
            case 39: return launch_gemm16<256, 128, 4, 2, 1, 3, false, true>(&a, st);
            case 40: return launch_gemm16<256, 256, 4, 2, 1, 2, false, true>(&a, st);
            case 42: return launch_gemm16<256, 256, 2, 4, 1, 8, false, true>(&a, st);
            case 45: return launch_gemm16<512, 128, 4, 2, 1, 8, false, true>(&a, st);
            case 48: return launch_gemm16<128, 80, 4, 1, 2, 3, false, true, false, 1, 32>(&a, st);
            case 49: return launch_gemm16<128, 160, 2, 2, 2, 2, false, true, false, 1, 64>(&a, st);
            case 50: return launch_gemm16<256, 160, 4, 2, 1, 3, false, true, false, 1, 32>(&a, st);
            case 51: return launch_gemm16<256, 160, 4, 2, 1, 2, false, true, false, 1, 64>(&a, st);
            default: return launch_gemm16<128, 80, 4, 1, 2, 3, false, true>(&a, st);
        }
    }
    const bool t = a.out_mode == 2;
    switch (tile) {
        case 39: return launch_gemm16<256, 128, 4, 2, 1, 3, false>(&a, st);
        case 40: return launch_gemm16<256, 256, 4, 2, 1, 2, false>(&a, st);
        case 42: return launch_gemm16<256, 256, 2, 4, 1, 8, false>(&a, st);
        case 45: return launch_gemm16<512, 128, 4, 2, 1, 8, false>(&a, st);
        case 38: return t ? launch_gemm16<128, 80, 4, 1, 1, 3, true>(&a, st) : launch_gemm16<128, 80, 4, 1, 1, 3, false>(&a, st);
        case 32: return t ? launch_gemm16<128, 80, 4, 1, 2, 2, true>(&a, st) : launch_gemm16<128, 80, 4, 1, 2, 2, false>(&a, st);
        case 33: return t ? launch_gemm16<128, 160, 2, 2, 2, 2, true>(&a, st) : launch_gemm16<128, 160, 2, 2, 2, 2, false>(&a, st);
        case 34:
            // knob 1 (tools only): the 256 x 160 tile with its eight waves as 4 x 2 (64 x 80 per wave: 18 fragment reads per K step
            // instead of 24) -- not for GEGLU (value / gate pairs need an even number of column fragments per wave)
            // default (knob 1 = 0): 4 x 2 for M >= 8192 (measured +2..4 % there, equal below); 1 forces 4 x 2, 2 forces 8 x 1
            if (a.act != 2 && (supir_debug_knob_value(1) == 1 || (supir_debug_knob_value(1) == 0 && a.M >= 8192)))
                return t ? launch_gemm16<256, 160, 4, 2, 1, 3, true>(&a, st) : launch_gemm16<256, 160, 4, 2, 1, 3, false>(&a, st);
            return t ? launch_gemm16<256, 160, 8, 1, 1, 3, true>(&a, st) : launch_gemm16<256, 160, 8, 1, 1, 3, false>(&a, st);
        default: return t ? launch_gemm16<128, 80, 4, 1, 2, 3, true>(&a, st) : launch_gemm16<128, 80, 4, 1, 2, 3, false>(&a, st);
    }
}

// Two problems in one launch (tiles 33 / 34 / 35; tile 32 is tile 35 with a shallower ring and has no grouped form).  What fixes the
// grid and the kernel instantiation must agree (M, N, K, output mode); every other argument -- pointers, strides, which optional
// operands are present, the convolution geometry -- is read per problem.
static bool g16_same_shape(const GemmArgs& x, const GemmArgs& y) {
    return x.M == y.M && x.N == y.N && x.K == y.K && x.out_mode == y.out_mode;
}

int supir_gemm16_launch_n(const GemmArgs* a, int n, hipStream_t st, int tile, bool conv) {
    if (n == 1) return supir_gemm16_launch(a[0], st, tile, conv);
    if (n != 2 || tile == 32 || tile >= 38) return SUPIR_ERR_SHAPE;      // (the halo tiles 48-51 have no grouped form)
    if (!supir_gemm16_supported(a[0], tile, conv) || !supir_gemm16_supported(a[1], tile, conv) || !g16_same_shape(a[0], a[1])) return SUPIR_ERR_SHAPE;
    // the wave arrangement of the 256 x 160 tile follows the single-launch policy (4 x 2 for convolutions and M >= 8192): a grouped launch
    // must stay bitwise the two single launches, GroupNorm partials and row statistics included (their cross-wave sums are ordered by it)
    const int knob = supir_debug_knob_value(1);
    if (conv) {
        switch (tile) {
            case 33: return launch_gemm16<128, 160, 2, 2, 2, 2, false, true, false, 2>(a, st);
            case 34: return knob != 2 ? launch_gemm16<256, 160, 4, 2, 1, 3, false, true, false, 2>(a, st)
                                      : launch_gemm16<256, 160, 8, 1, 1, 3, false, true, false, 2>(a, st);
            default: return launch_gemm16<128, 80, 4, 1, 2, 3, false, true, false, 2>(a, st);
        }
    }
    const bool t = a[0].out_mode == 2;
    const bool w42 = a[0].act != 2 && (knob == 1 || (knob == 0 && a[0].M >= 8192));
    switch (tile) {
        case 33: return t ? launch_gemm16<128, 160, 2, 2, 2, 2, true, false, false, 2>(a, st) : launch_gemm16<128, 160, 2, 2, 2, 2, false, false, false, 2>(a, st);
        case 34:
            if (w42) return t ? launch_gemm16<256, 160, 4, 2, 1, 3, true, false, false, 2>(a, st) : launch_gemm16<256, 160, 4, 2, 1, 3, false, false, false, 2>(a, st);
            return t ? launch_gemm16<256, 160, 8, 1, 1, 3, true, false, false, 2>(a, st) : launch_gemm16<256, 160, 8, 1, 1, 3, false, false, false, 2>(a, st);
        default: return t ? launch_gemm16<128, 80, 4, 1, 2, 3, true, false, false, 2>(a, st) : launch_gemm16<128, 80, 4, 1, 2, 3, false, false, false, 2>(a, st);
    }
}

// fused q|k|v projection: columns [0, n_split) -> C (normal epilogue), [n_split, N) -> C2 transposed.  Two tile widths, 256 x 160 (tile 34's
// loop, eight waves as 8 x 1) and 256 x 128 (tile 39's, 4 x 2): the launch takes the one with fewer rounds-of-256-workgroups x columns.
// (2048, 3840, 1280) -- the 1280-wide SpatialTransformers at 32^2 -- is 192 tiles of 256 x 160 (64 CUs idle) or 240 of 256 x 128, each
// 0.8 x the work: one round either way.  Same K order per output element: bitwise-equal results.
static int g16_qkv_check(const GemmArgs& a, int bn) {
    if (a.M % 256 || a.N % bn || a.n_split % bn || a.n_split <= 0 || a.n_split >= a.N || a.K % 64 || (a.K >> 6) < 2) return SUPIR_ERR_SHAPE;
    if (a.lda % 8 || a.ldc % 8 || (((size_t)a.C) & 15) || a.ldc2 % 4 || a.rows_per_batch % 4 || a.ln_slots > 32) return SUPIR_ERR_SHAPE;
    if (a.act != 0 || a.out_mode != 0 || a.res || a.rowbias || a.rowstats_out || !a.C2) return SUPIR_ERR_ARG;
    return SUPIR_OK;
}

// knob 4 (tools only): 1 forces 160 columns, 2 forces 128 where both fit
static int g16_qkv_bn(const GemmArgs& a) {
    const bool ok160 = a.N % 160 == 0 && a.n_split % 160 == 0, ok128 = a.N % 128 == 0 && a.n_split % 128 == 0;
    if (!ok128 || !ok160) return ok128 ? 128 : 160;
    const int knob = supir_debug_knob_value(4);
    if (knob == 1 || knob == 2) return knob == 1 ? 160 : 128;
    const long tm = a.M / 256;
    const long c160 = ((tm * (a.N / 160) + 255) / 256) * 160, c128 = ((tm * (a.N / 128) + 255) / 256) * 128;
    return c128 < c160 ? 128 : 160;
}

int supir_gemm16_qkv_launch(const GemmArgs& a, hipStream_t st) {
    const int bn = g16_qkv_bn(a);
    const int rc = g16_qkv_check(a, bn);
    if (rc != SUPIR_OK) return rc;
    if (bn == 128) return launch_gemm16<256, 128, 4, 2, 1, 3, false, false, true>(&a, st);
    if (supir_debug_knob_value(1)) return launch_gemm16<256, 160, 4, 2, 1, 3, false, false, true>(&a, st);
    return launch_gemm16<256, 160, 8, 1, 1, 3, false, false, true>(&a, st);
}

int supir_gemm16_qkv_launch_n(const GemmArgs* a, int n, hipStream_t st) {
    if (n == 1) return supir_gemm16_qkv_launch(a[0], st);
    if (n != 2) return SUPIR_ERR_SHAPE;
    for (int q = 0; q < 2; ++q) {
        const int rc = g16_qkv_check(a[q], 160);
        if (rc != SUPIR_OK) return rc;
    }
    if (!g16_same_shape(a[0], a[1])) return SUPIR_ERR_SHAPE;
    return launch_gemm16<256, 160, 8, 1, 1, 3, false, false, true, 2>(a, st);
}
