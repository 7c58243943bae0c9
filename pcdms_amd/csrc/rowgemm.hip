// Thin-K GEMM with the ACTIVATION operand stationary in registers (K = 320: the level-0 Transformer2D / BasicTransformerBlock linears,
// /root/reference/src/models/stage2_inpaint_unet_2d_condition.py:321-344,407-430 -> diffusers Attention to_q/k/v, FeedForward GEGLU,
// proj_in / proj_out).
//
//   out[m, n] = epilogue( sum_k LN?(A)[m, k] * W[n, k] ),  M = 45056 / 22528 rows, K = 320, N = 320 .. 2560.
//
// Why a second GEMM kernel: at K = 320 a tiled GEMM has five K-tiles per output tile -- the tile loop is all prologue and epilogue
// (tools/gemm_anatomy.py: 13.8 k cycles of K loop against 4.5 k + 10-22 k of prologue / epilogue per tile), every 128x128 output tile
// re-stages its A rows, and the LayerNorm in front of to_q|k|v / to_q / GEGLU is a launch of its own (one read + one write of the
// tensor).  Here:
//  * one workgroup = WGM x WGN waves; a wave owns FMW*16 = 48 rows for the whole launch and keeps them as MFMA operand fragments
//    in registers (16x16x32: 10 k-steps x 3 row blocks x 4 VGPRs = 120); the A rows are read straight into registers.  The instance the
//    product uses (tile 34) has FOUR waves and runs TWO workgroups per CU (__launch_bounds__(256, 2), 80 KB of LDS each, the N tiles
//    split over gridDim.y): one wave of each workgroup per SIMD = two barrier domains per CU, so that one workgroup's stage barrier,
//    DMA issue and epilogue sit under the other's MFMAs.  The 8-wave one-workgroup-per-CU instances (tiles 31-33) never beat the tiled
//    kernel (tools/bench_rowgemm.py, profiles/r3_bench_rowgemm.txt);
//  * LayerNorm (optional, template LNF) is FOLDED: gamma goes into the weights and beta into the bias when they are packed
//    (W' = W diag(gamma), b' = b + W beta), and since LN(x) W'^T = rstd (x W'^T - mean * rowsum(W')), all that is left at run time is
//    a per-row mean / rstd -- taken from the registers right after the load (a row is spread over 4 lanes: local sums, two
//    xor-shuffles, exact two-pass variance) -- and one fused multiply-add per accumulator in the epilogue: out = rstd (acc - mean
//    wsum[n]) + b'[n].  The normalised rows never exist, not even in registers (the first version normalised the fragments in place:
//    ~2000 VALU instructions per wave in the prologue, 15 k cycles -- as long as the LayerNorm launch it replaced);
//  * the workgroup then walks over ALL N tiles (BN = 64 / 128 columns): W tiles stream HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds, as
//    gemm.hip) through ONE ring of NSTG stages of 64 k that runs across N-tile boundaries: the next tile's first stages are in flight
//    while the current tile's epilogue runs, 5-7 stages (40-96 KiB) outstanding per CU; only B fragments are read from LDS
//    (FN ds_read_b128 per FN*FMW MFMAs);
//  * M = 45056 gives 235 row blocks of 192 rows; tile 34 splits the N tiles of a row block over 512 / 235 = 2 workgroups (470 in
//    flight, two per CU), M = 22528 over 4;
//  * epilogues: bias (+ residual) store, GEGLU ([32 h | 32 gate] packed rows, as gemm.hip), q|k store + V^T transposed store -- all
//    through a wave-private fp32 LDS tile so that every global access is 16 B per lane along the contiguous axis.
// vmcnt bookkeeping: vector-memory operations of a wave retire from its counter in issue order, loads and stores alike (gfx9 has one
// counter for both and hipcc's own waits rely on that order).  The ring's counted waits therefore count, besides the younger ring
// stages, the epilogue STORES issued since the awaited stage (a fixed number per N tile); loads the epilogue issues (bias, residual)
// are not counted: an uncounted operation can only make a wait longer than necessary, never shorter.  The first version counted ring
// stages only: every wait that followed an epilogue then also waited for that epilogue's store acknowledgements (~1.5 us per N tile).
// In the steady state the count depends only on the K-tile index inside the N tile: a compile-time s_waitcnt (wait_stage_const); the
// run-time form (a switch over immediates: hipcc lowers it to ~35 scalar compares and branches, ~280 cycles per stage) is left to the
// first and last stages of a launch.
#include "gemm_args.h"

#include <type_traits>

namespace {
using pcdm_gemm_detail::GemmArgs;

constexpr int kK = 320;            // the contraction length this kernel is built for
constexpr int kKS = kK / 32;       // MFMA k-steps (16x16x32)
constexpr int kNKT = kK / 64;      // ring stages per N tile
constexpr uint32_t kOOB = 0x80000000u;

// keeps a packed bf16x8 register value opaque to the optimiser: without it the fp32 conversions of the LayerNorm passes are hoisted
// and kept live side by side (3 x 80 floats per lane on top of the packed rows): spills
#ifdef PCDM_EMU
#define RG_KEEP_PACKED(v) ((void)0)
#else
#define RG_KEEP_PACKED(v) asm volatile("" : "+v"(v))
#endif

// s_waitcnt vmcnt(n) + lgkmcnt(0) (the stage barrier follows); n is wave-uniform at run time (a scalar switch over immediates)
__device__ __forceinline__ void wait_stage(int n) {
#ifdef PCDM_EMU
    (void)n;
#else
    __builtin_amdgcn_sched_barrier(0);
    // simm16: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
#define PCDM_RG_W(N) case N: __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14)); break;
    switch (n) {
        PCDM_RG_W(0) PCDM_RG_W(1) PCDM_RG_W(2) PCDM_RG_W(3) PCDM_RG_W(4) PCDM_RG_W(5) PCDM_RG_W(6) PCDM_RG_W(7) PCDM_RG_W(8)
        PCDM_RG_W(9) PCDM_RG_W(10) PCDM_RG_W(11) PCDM_RG_W(12) PCDM_RG_W(13) PCDM_RG_W(14) PCDM_RG_W(15) PCDM_RG_W(16)
        PCDM_RG_W(17) PCDM_RG_W(18) PCDM_RG_W(19) PCDM_RG_W(20) PCDM_RG_W(21) PCDM_RG_W(22) PCDM_RG_W(23) PCDM_RG_W(24) PCDM_RG_W(25)
        PCDM_RG_W(26) PCDM_RG_W(27) PCDM_RG_W(28) PCDM_RG_W(29) PCDM_RG_W(30) PCDM_RG_W(31) PCDM_RG_W(32)
        default: __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8) | (0 << 14)); break;
    }
#undef PCDM_RG_W
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// the same wait with a compile-time count: ONE instruction.  (The run-time form above compiles to a tree of ~35 scalar compares and
// branches through hipcc's structured-control-flow lowering -- a few hundred cycles per ring stage; the steady state of the ring, whose
// count depends only on the K-tile index inside the N tile, therefore uses this one.)
template <int N>
__device__ __forceinline__ void wait_stage_const() {
#ifndef PCDM_EMU
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ void stage_barrier() {
#ifdef PCDM_EMU
    __syncthreads();
#else
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#endif
}

template <int N>
__device__ __forceinline__ void wait_lds_keep() {   // s_waitcnt lgkmcnt(N): all but this wave's N most recent LDS operations are complete
#ifndef PCDM_EMU
    __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (N << 8) | (3 << 14));
#endif
}

__device__ __forceinline__ void wait_lds_rg() {   // s_waitcnt lgkmcnt(0), visible to the compiler's scoreboard
#ifndef PCDM_EMU
    __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));
#endif
}

// WGM x WGN waves; a wave owns FMW row blocks of 16 and, per N tile, BN / WGN columns; NSTG ring stages of [BN][64]; LNF: folded LayerNorm;
// GLU: the gated-linear-unit epilogue (its own instantiation: with all three epilogues in one kernel the lane constants hipcc hoists out of
// the N-tile loop for each of them pushed the 48 x 64 wave tile over 256 VGPRs -- spills whose reloads sit behind the epilogue's stores)
// NW = 4 (256 threads, __launch_bounds__(256, 2)): TWO workgroups per CU, one wave of each per SIMD -- two barrier domains, so that one
// workgroup's stage wait / barrier / DMA issue can sit under the other's MFMAs.  Such a launch splits the N tiles over gridDim.y
// workgroups per row block (there are only 235 row blocks of 192 at M = 45056; the A rows are then read once per N split).
template <int WGM, int WGN, int FMW, int BN, int NSTG, bool LNF, bool GLU>
__global__ __launch_bounds__(WGM* WGN * 64, WGM* WGN == 4 ? 2 : 1) void rowgemm_kernel(const GemmArgs p) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * FMW * 16;
    constexpr int WNC = BN / WGN;          // columns of an N tile per wave
    constexpr int FN = WNC / 16;           // B fragments per wave per k-step
    constexpr int DPW = BN / 8 / NW;       // LDS-DMA instructions per wave per stage (8 rows of 128 B each)
    constexpr int EPW = WNC + 4;           // fp32 pitch of the wave-private epilogue tile (16 rows)
    static_assert((NW == 8 || NW == 4) && BN % (8 * NW) == 0 && WNC % 16 == 0 && (WNC == 32 || WNC == 64), "shape");
    static_assert(NSTG >= 3 && (NSTG - 2) * DPW <= 14, "ring depth");   // (stages q + 1 .. q + NSTG - 2 in flight behind the one awaited)
    PCDM_DYN_SMEM(smem);
    u16* Ws = (u16*)smem;                                  // [NSTG][BN][64]  (unpadded, 16-byte chunks XOR-swizzled by (row >> 1) & 7)
    float* eps_all = (float*)(Ws + NSTG * BN * 64);        // [NW][16][EPW]
    float* bias_s = eps_all + NW * 16 * EPW;               // [Npad]: the epilogue reads no global memory (a load behind a store would
                                                           // wait for that store's acknowledgement: vmcnt retires in issue order)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
#ifndef PCDM_EMU
    // p.debug & 4 (tools/rowgemm_anatomy.py): per-wave cycle accounting into ws[(workgroup, wave)][8] (uint64)
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool stamps = (p.debug & 4) != 0;
#define RG_T0() do { if (stamps) tprev = __builtin_readcyclecounter(); } while (0)
#define RG_ACC(i) do { if (stamps) { const unsigned long long n_ = __builtin_readcyclecounter(); tk[i] += n_ - tprev; tprev = n_; } } while (0)
#else
#define RG_T0() ((void)0)
#define RG_ACC(i) ((void)0)
#endif
    RG_T0();
    const int m0 = blockIdx.x * BM + wm * (FMW * 16);      // first row of this wave
    const int lrow = lane & 15, lq = lane >> 4;            // fragment row / 8-element k chunk of the lane
    float* ep = eps_all + wave * (16 * EPW);

    // ---- W ring: stage q = (N tile q / kNKT, K tile q % kNKT).  Per-lane offset constant over the launch, SGPR offset per stage.
    const BufRsrc rs_w = make_buf_rsrc(p.w);
    uint32_t b_off[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int rl = (wave * DPW + i) * 8 + (lane >> 3);
        b_off[i] = (uint32_t)((int64_t)rl * p.ldw * 2) + (uint32_t)(((lane & 7) ^ ((rl >> 1) & 7)) * 16);
    }
    // this workgroup's N tiles: [nt_lo, nt_lo + NT) of the Npad / BN tiles (gridDim.y == 1: all of them)
    const int NT_all = p.Npad / BN;
    const int nt_lo = (int)((int64_t)blockIdx.y * NT_all / gridDim.y);
    const int NT = (int)((int64_t)(blockIdx.y + 1) * NT_all / gridDim.y) - nt_lo;
    // Every workgroup walks ALL N tiles, i.e. streams the whole of W -- and with one workgroup per CU starting together they would all ask
    // the L2s for the SAME 16 KiB at the same moment, stage after stage (one channel serving a line 32 times over while the others idle).
    // Each workgroup therefore starts its walk at a different N tile (p.debug & 16: all start at tile 0, for A/B runs).
    const int rot = (p.debug & 16) ? 0 : (int)((blockIdx.x * 7u) % (unsigned)NT);
    float* wsum_s = bias_s + p.Npad;                       // [Npad] row sums of the gamma-folded weights (LNF)
    float* stat_w = wsum_s + p.Npad + wave * (FMW * 16 * 2);   // [FMW * 16][2] {mean, rstd} of this wave's rows (LNF; kept out of registers)
    for (int i = t * 4; i < p.Npad; i += NW * 64 * 4) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(bias_s + i) = p.bias ? *(const f32x4*)(p.bias + i) : z4;
        if constexpr (LNF) *(f32x4*)(wsum_s + i) = *(const f32x4*)(p.ln_wsum + i);
    }
    const bool skip = m0 - wm * (FMW * 16) + BM <= p.zero_rows;   // every row of the workgroup is declared zero: epilogue only
    const int Q = skip ? 0 : NT * kNKT;
    auto issue = [&](int q) {
        const int nt_ = q / kNKT, kt = q - nt_ * kNKT;
        const int nt = nt_lo + (nt_ + rot < NT ? nt_ + rot : nt_ + rot - NT);
        const uint32_t soff = (uint32_t)(((int64_t)nt * BN * p.ldw + kt * 64) * 2);
        u16* ws = Ws + (q % NSTG) * (BN * 64) + (wave * DPW) * 8 * 64;
#pragma unroll
        for (int i = 0; i < DPW; ++i) buf_glds16(rs_w, b_off[i], soff, ws + i * 8 * 64);
    };
#pragma unroll
    for (int s = 0; s < NSTG - 1; ++s)
        if (s < Q) issue(s);

    // ---- A rows -> MFMA operand fragments in registers: xa[j][ks] = row (16 j + lane % 16), k = 32 ks + 8 (lane / 16) .. + 7
    u16x8 xa[FMW][kKS];
    {
        const BufRsrc rs_a = make_buf_rsrc(p.a);
#pragma unroll
        for (int j = 0; j < FMW; ++j) {
            const int m = m0 + j * 16 + lrow;
            const uint32_t v0 = (m < p.M && m >= p.zero_rows) ? (uint32_t)(((int64_t)m * p.lda + 8 * lq) * 2) : kOOB;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) xa[j][ks] = __builtin_bit_cast(u16x8, buf_load16(rs_a, v0 == kOOB ? kOOB : v0 + ks * 64));
        }
    }
    // LayerNorm statistics of each row (4 lanes x 80 elements): mean, then the centred sum of squares (exact two-pass, fp32).  Lane
    // (lrow, lq) ends up with the statistics of row 16 j + lrow -- the pixel row whose accumulators it holds in the epilogue.
    if constexpr (LNF) {
#pragma unroll
        for (int j = 0; j < FMW; ++j) {
            float mean_j, rstd_j;
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += bf2f(xa[j][ks][e]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            mean_j = s * (1.0f / kK);
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) RG_KEEP_PACKED(xa[j][ks]);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = bf2f(xa[j][ks][e]) - mean_j;
                    q += d * d;
                }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            rstd_j = 1.0f / sqrtf(q * (1.0f / kK) + p.ln_eps);
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) RG_KEEP_PACKED(xa[j][ks]);
            if (lq == 0) {
                stat_w[(j * 16 + lrow) * 2] = mean_j;
                stat_w[(j * 16 + lrow) * 2 + 1] = rstd_j;
            }
        }
    }

    RG_ACC(0);   // prologue: ring start, A rows, LayerNorm
    // ---- epilogue operands
    constexpr bool geglu = GLU;
    const int vt0 = p.epilogue == PCDM_EPI_SPLIT_VT ? p.vt_col0 : 0x7fffffff;   // columns >= vt0 go to out2 transposed
    const int ncols_out = p.epilogue == PCDM_EPI_SPLIT_VT ? p.vt_col0 : (geglu ? p.N : p.N);
    const bool has_res = p.residual != nullptr && !geglu;
    // (p.debug & 32, tools/rowgemm_anatomy.py: a zero-size output descriptor -- every store is issued and dropped)
    const BufRsrc rs_o = make_buf_rsrc(p.out, (p.debug & 32) ? 0u : (uint32_t)((((int64_t)p.M - 1) * p.ldo + ncols_out) * 2));
    const BufRsrc rs_r = make_buf_rsrc(has_res ? (const void*)p.residual : (const void*)p.out,
                                       has_res ? (uint32_t)((((int64_t)p.M - 1) * p.ldr + p.N) * 2) : 0u);
    const BufRsrc rs_vt = make_buf_rsrc(p.out2 ? (const void*)p.out2 : (const void*)p.out,
                                        p.out2 ? (uint32_t)((int64_t)(p.M / p.rows_per_batch) * (p.N - p.vt_col0) * p.ldo2 * 2) : 0u);

    f32x4 acc[FN][FMW];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FMW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // One N tile's results: lane holds, per (i, j), channels 16 i + 4 lq + {0..3} of pixel row 16 j + lrow.  Row block by row block:
    // quads -> ep[row][channel] (fp32) -> read back along the contiguous axis, 8 channels (or 8 tokens) per lane.
    // accumulator quad (i, j) with the folded LayerNorm applied: rstd (acc - mean wsum[n])
    auto ln_quad = [&](const f32x4& a, int n_quad, int j) -> f32x4 {
        if constexpr (LNF) {
            const f32x4 w4 = *(const f32x4*)(wsum_s + n_quad);
            const float mean_j = stat_w[(j * 16 + lrow) * 2], rstd_j = stat_w[(j * 16 + lrow) * 2 + 1];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rstd_j * (a[e] - mean_j * w4[e]);
            return v;
        } else {
            return a;
        }
    };
    auto epilogue = [&](int nt_) {
        const int nt = nt_lo + (nt_ + rot < NT ? nt_ + rot : nt_ + rot - NT);
        const int n0w = nt * BN + wn * WNC;                 // first (packed) column of this wave's tile
        if constexpr (GLU) {
            if constexpr (WNC == 64) {
                // packed rows alternate [32 h | 32 gate]: fragments 0, 1 = h, 2, 3 = gate of output channels n0w / 2 .. + 31
                const bool swiglu = p.act == PCDM_ACT_SILU;   // (wave-uniform: one branch per row block, not one per element)
                const int no = n0w / 2 + (lane & 3) * 8;    // read-back: 4 lanes per row, 16 rows per instruction
                const int rr = lane >> 2;
#pragma unroll
                for (int j = 0; j < FMW; ++j) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x4 v;
                        const int nh = n0w + i * 16 + 4 * lq;      // packed column of h; its gate is 32 further (biases re-read from LDS
                        const f32x4 ah = ln_quad(acc[i][j], nh, j) + *(const f32x4*)(bias_s + nh);        // per quad: registers are scarce here)
                        const f32x4 ag = ln_quad(acc[i + 2][j], nh + 32, j) + *(const f32x4*)(bias_s + nh + 32);
                        if (swiglu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = ah[e] * silu_f(ag[e]);
                        } else {
                            v = geglu_quad(ah, ag);
                        }
                        *(f32x4*)(ep + lrow * EPW + i * 16 + 4 * lq) = v;
                    }
                    PCDM_WAVE_SYNC();
                    const f32x4 v0 = *(const f32x4*)(ep + rr * EPW + (lane & 3) * 8), v1 = *(const f32x4*)(ep + rr * EPW + (lane & 3) * 8 + 4);
                    u32x4 o = {pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]), pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3])};
                    const int m = m0 + j * 16 + rr;
                    buf_store16(rs_o, no < p.N ? (uint32_t)(((int64_t)m * p.ldo + no) * 2) : kOOB, o);
                    PCDM_WAVE_SYNC();
                }
            }
            return;
        } else {
        if (n0w >= vt0) {
            // V^T: a lane takes one channel and 8 consecutive tokens; out2[b, channel, token]
#pragma unroll
            for (int j = 0; j < FMW; ++j) {
                const int mb = m0 + j * 16;                 // 16 rows inside one batch entry (rows_per_batch % 16 == 0)
                const int b = mb / p.rows_per_batch, tok0 = mb - b * p.rows_per_batch;
#pragma unroll
                for (int i = 0; i < FN; ++i) *(f32x4*)(ep + lrow * EPW + i * 16 + 4 * lq) = ln_quad(acc[i][j], n0w + i * 16 + 4 * lq, j);
                PCDM_WAVE_SYNC();
#pragma unroll
                for (int ii = 0; ii < WNC / 32; ++ii) {
                    const int c = ii * 32 + (lane & 31), th = lane >> 5, n = n0w + c;
                    const float bias = bias_s[n];
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ep[(8 * th + e) * EPW + c] + bias;
                    u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
                    const uint32_t vo = (n < p.N && mb < p.M)
                                            ? (uint32_t)((((int64_t)b * (p.N - p.vt_col0) + (n - p.vt_col0)) * p.ldo2 + tok0 + 8 * th) * 2)
                                            : kOOB;
                    buf_store16(rs_vt, vo, o);
                }
                PCDM_WAVE_SYNC();
            }
            return;
        }
        // plain rows: out = acc + bias (+ residual), 8 channels per lane
        constexpr int LPR = WNC / 8, RPI = 64 / LPR, NIT = 16 / RPI;
        const int rl = lane / LPR, c8 = (lane - rl * LPR) * 8;
        const int n = n0w + c8;
        const bool nok = n < p.N;
        const f32x4 b0 = *(const f32x4*)(bias_s + n), b1 = *(const f32x4*)(bias_s + n + 4);   // (n + 7 < Npad)
        // every residual row of the tile is requested BEFORE its first store: the counter retires in issue order, so a load issued
        // behind a store returns only after that store's acknowledgement (a descriptor of size 0 -- no residual -- returns zeros
        // without touching memory)
        u32x4 rv[FMW][NIT];
#pragma unroll
        for (int j = 0; j < FMW; ++j)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int m = m0 + j * 16 + it * RPI + rl;
                rv[j][it] = buf_load16_once(rs_r, nok ? (uint32_t)(((int64_t)m * p.ldr + n) * 2) : kOOB);
            }
#pragma unroll
        for (int j = 0; j < FMW; ++j) {
#pragma unroll
            for (int i = 0; i < FN; ++i) *(f32x4*)(ep + lrow * EPW + i * 16 + 4 * lq) = ln_quad(acc[i][j], n0w + i * 16 + 4 * lq, j);
            PCDM_WAVE_SYNC();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int r = it * RPI + rl;
                f32x4 a0 = *(const f32x4*)(ep + r * EPW + c8) + b0, a1 = *(const f32x4*)(ep + r * EPW + c8 + 4) + b1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0[e] += __builtin_bit_cast(float, rv[j][it][e >> 1] << (e & 1 ? 0 : 16) & 0xffff0000u);
                    a1[e] += __builtin_bit_cast(float, rv[j][it][2 + (e >> 1)] << (e & 1 ? 0 : 16) & 0xffff0000u);
                }
                u32x4 o = {pack2bf(a0[0], a0[1]), pack2bf(a0[2], a0[3]), pack2bf(a1[0], a1[1]), pack2bf(a1[2], a1[3])};
                const int m = m0 + j * 16 + r;
                buf_store16(rs_o, nok ? (uint32_t)(((int64_t)m * p.ldo + n) * 2) : kOOB, o);
            }
            PCDM_WAVE_SYNC();
        }
        }
    };

    // stores every epilogue is guaranteed to issue per wave (masked lanes / rows beyond M still issue the instruction)
    const int n_ep_stores = geglu ? FMW : FMW * (WNC / 32);
    if (skip) {   // all rows zero: out = bias (+ residual)
        __syncthreads();   // (bias_s)
        for (int nt = 0; nt < NT; ++nt) epilogue(nt);
        return;
    }

    // ---- main loop.  Iteration q: [wait: stages <= q + 1 landed for the whole workgroup, everybody done reading stage q - 1] -> refill the
    // freed slot with stage q + NSTG - 1 -> 2 k-steps of MFMAs from registers (A) x LDS (W).  The B fragments of a k-step are requested one
    // k-step ahead, into the other of two register buffers (also across the stage boundary: that is what the one-stage-ahead wait is for),
    // so their LDS latency is covered by the MFMAs in front of it -- in the first version every k-step waited out its own reads with
    // the matrix pipe idle (tools/rowgemm_anatomy.py: 1450 cycles per stage for 770 cycles of MFMA issue on the SIMD).  Only the first
    // k-step of an N tile, behind the epilogue, is exposed.  The N-tile loop is a run-time loop, its 5 K-tiles are unrolled (the A
    // fragments are indexed by compile-time k-step numbers: registers).
    const int frow_sw = (lrow >> 1) & 7;
    u16x8 wf[2][FN];
    auto load_wf = [&](int q, int k2, auto bsel) {
        constexpr int bb = decltype(bsel)::value;
        const u16* ws = Ws + (q % NSTG) * (BN * 64) + (wn * WNC + lrow) * 64 + ((k2 * 4 + lq) ^ frow_sw) * 8;
#pragma unroll
        for (int i = 0; i < FN; ++i) wf[bb][i] = *(const u16x8*)(ws + i * 16 * 64);
    };
    auto mfma_step = [&](int ks, auto bsel) {
        constexpr int bb = decltype(bsel)::value;
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FMW; ++j) acc[i][j] = mfma_16x16x32(wf[bb][i], xa[j][ks], acc[i][j]);
    };
    typedef std::integral_constant<int, 0> B0;
    typedef std::integral_constant<int, 1> B1;
    constexpr int EP_STORES = GLU ? FMW : FMW * (WNC / 32);   // (= n_ep_stores)
    for (int nt = 0; nt < NT; ++nt) {
        auto stage = [&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            const int q = nt * kNKT + kt;
            // awaited stage: q + 1 (q itself at the very end).  Operations this wave may leave in flight: the ring stages younger than
            // that, and the stores of the epilogues it ran since it issued the awaited stage (in iteration lo = target - NSTG + 1, or in
            // the prologue): an iteration i ends with an epilogue iff i % kNKT == kNKT - 1: q / kNKT - lo / kNKT of them in [lo, q - 1]
            const int tgt = q + 1 < Q ? q + 1 : q;
            const int younger = (Q - 1 - tgt) < (NSTG - 2 - (tgt - q)) ? (Q - 1 - tgt) : (NSTG - 2 - (tgt - q));
            const int lo = tgt - (NSTG - 1) > 0 ? tgt - (NSTG - 1) : 0;
            RG_ACC(4);
            // steady state (NSTG - 3 younger stages exist, and the epilogues between the issue of the awaited stage -- iteration
            // lo = q + 2 - NSTG -- and now have all happened): younger == NSTG - 3 and q / kNKT - lo / kNKT == ceil((NSTG - 2 - kt) / kNKT), a
            // compile-time count
            constexpr int NEP = NSTG - 2 - kt > 0 ? (NSTG - 2 - kt + kNKT - 1) / kNKT : 0;
            if (nt >= NEP && q + NSTG - 1 <= Q && !(p.debug & 128))
                wait_stage_const<(NSTG - 3) * DPW + NEP * EP_STORES>();
            else
                wait_stage(younger * DPW + (q / kNKT - lo / kNKT) * n_ep_stores);   // (p.debug & 128: always this form, for A/B runs)
            RG_ACC(1);   // waiting for this wave's part of the stage to land
            stage_barrier();
            RG_ACC(5);   // waiting for the other waves
            // The two waves of a SIMD (w and w + 4) place the refill at opposite ends of the iteration: an LDS-DMA instruction costs its wave
            // ~175 cycles of issue (tools/rowgemm_anatomy.py: 350 per stage with both waves doing it right behind the barrier, the matrix
            // pipe idle), so one wave issues while its partner multiplies, then they swap (p.debug & 64: both in front, for A/B runs)
            const bool dma_first = wave < NW / 2 || (p.debug & 64);
            if (dma_first && q + NSTG - 1 < Q) issue(q + NSTG - 1);
            RG_ACC(4);   // DMA issue
            if (kt == 0) load_wf(q, 0, B0());            // (kt > 0: requested during the previous stage's second k-step)
            // k-step 2 kt: request the fragments of k-step 2 kt + 1 (buffer 1), wait for buffer 0 only, multiply
            load_wf(q, 1, B1());
            wait_lds_keep<FN>();
            PCDM_SCHED_BARRIER();
            mfma_step(kt * 2, B0());
            PCDM_SCHED_BARRIER();
            // k-step 2 kt + 1: request the first fragments of the next stage (same N tile only), wait for buffer 1, multiply
            if (kt + 1 < kNKT) {
                load_wf(q + 1, 0, B0());
                wait_lds_keep<FN>();
            } else {
                wait_lds_rg();
            }
            PCDM_SCHED_BARRIER();
            mfma_step(kt * 2 + 1, B1());
            PCDM_SCHED_BARRIER();
            if (!dma_first && q + NSTG - 1 < Q) issue(q + NSTG - 1);
            RG_ACC(2);   // fragment reads + MFMAs (+ the late DMA issue)
        };
        static_assert(kNKT == 5, "K = 320");
        stage(std::integral_constant<int, 0>());
        stage(std::integral_constant<int, 1>());
        stage(std::integral_constant<int, 2>());
        stage(std::integral_constant<int, 3>());
        stage(std::integral_constant<int, 4>());
        epilogue(nt);
        zero_acc();
        RG_ACC(3);       // epilogue
    }
#ifndef PCDM_EMU
    if (stamps && p.ws && lane == 0) {
        unsigned long long* o = (unsigned long long*)p.ws + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = tk[i];
    }
#endif
}

template <int WGM, int WGN, int FMW, int BN, int NSTG, int NSPLIT_WGS = 0>
int launch_rg(const GemmArgs& a, hipStream_t st) {
    constexpr int BM = WGM * FMW * 16, WNC = BN / WGN, NW = WGM * WGN;
    constexpr int smem_fixed = NSTG * BN * 64 * (int)sizeof(u16) + NW * 16 * (WNC + 4) * (int)sizeof(float);
    if (a.Npad % BN || a.Npad > 2560) return -1;
    const int smem = smem_fixed + (2 * a.Npad + NW * FMW * 32) * (int)sizeof(float);   // + bias, the folded weights' row sums, row statistics
    if (a.epilogue == PCDM_EPI_GEGLU && WNC != 64) return -1;
    if (a.epilogue == PCDM_EPI_SPLIT_VT && (a.vt_col0 % WNC || a.rows_per_batch % 16 || (a.ldo2 & 7) || a.M % 16)) return -1;
    const int gx = (a.M + BM - 1) / BM;
    // NSPLIT_WGS > 0: split the N tiles over enough workgroups per row block to reach that many workgroups in total (two per CU)
    int gy = 1;
    if (NSPLIT_WGS > 0) {
        gy = (NSPLIT_WGS + gx / 2) / gx;
        gy = gy < 1 ? 1 : (gy > a.Npad / BN ? a.Npad / BN : gy);
    }
    const dim3 grid(gx, gy);
    const int smem_max = smem_fixed + (2 * 2560 + NW * FMW * 32) * (int)sizeof(float);
#define PCDM_RG_LAUNCH(LN_, GLU_)                                                                                                      \
    do {                                                                                                                               \
        static bool attr_done = false;                                                                                                 \
        if (!attr_done) {                                                                                                              \
            (void)hipFuncSetAttribute((const void*)rowgemm_kernel<WGM, WGN, FMW, BN, NSTG, LN_, GLU_>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem_max);                                          \
            attr_done = true;                                                                                                          \
        }                                                                                                                              \
        PCDM_LAUNCH(PCDM_KERNEL_NAME(rowgemm_kernel<WGM, WGN, FMW, BN, NSTG, LN_, GLU_>), grid, dim3(NW * 64), smem, st, a);     \
    } while (0)
    const bool glu = a.epilogue == PCDM_EPI_GEGLU;
    if constexpr (WNC == 64) {
        if (a.ln_wsum && glu) PCDM_RG_LAUNCH(true, true);
        else if (a.ln_wsum) PCDM_RG_LAUNCH(true, false);
        else if (glu) PCDM_RG_LAUNCH(false, true);
        else PCDM_RG_LAUNCH(false, false);
    } else {
        if (a.ln_wsum) PCDM_RG_LAUNCH(true, false);
        else PCDM_RG_LAUNCH(false, false);
    }
#undef PCDM_RG_LAUNCH
    PCDM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

// Tile ids 31.. of pcdm_gemm_params.tile.  Takes: linear, single source, K = 320, no split-K, no row vector, no activation
// (GEGLU's gate activation excepted), epilogues STORE / GEGLU / SPLIT_VT, 16-byte aligned rows.
int pcdm_gemm_detail::launch_rowgemm(int tile, const GemmArgs& a, hipStream_t st) {
    if (a.K != kK || a.a2 || a.split_k > 1 || a.rowvec || (a.act && a.epilogue != PCDM_EPI_GEGLU)) return -1;
    if (a.epilogue != PCDM_EPI_STORE && a.epilogue != PCDM_EPI_GEGLU && a.epilogue != PCDM_EPI_SPLIT_VT) return -1;
    if ((a.lda & 7) || (a.ldo & 7) || (a.N & 7) || (a.ldw & 7)) return -1;
    if (a.residual && ((a.ldr & 7) || a.res_mod < a.M)) return -1;
    if (a.epilogue == PCDM_EPI_GEGLU && (!a.bias || a.residual)) return -1;
    if (a.ln_wsum && a.zero_rows) return -1;
    switch (tile) {
        case 31: return launch_rg<4, 2, 3, 128, 6>(a, st);   // 192 rows, N tiles of 128 (waves 48 x 64: GEGLU-capable), 96 + 34 KiB
        case 32: return launch_rg<4, 2, 3, 64, 8>(a, st);    // 192 rows, N tiles of 64 (waves 48 x 32), 64 + 18 KiB
        case 33: return launch_rg<2, 4, 3, 128, 6>(a, st);   // 96 rows, N tiles of 128 (waves 48 x 32): M = 22528 -> 235 workgroups
        case 34: return launch_rg<4, 1, 3, 64, 5, 512>(a, st);  // FOUR waves (48 x 64 each), 192 rows, N tiles of 64, two workgroups per CU
                                                               // (40 + 17 + <= 22 KiB each), N split over 512 / row-blocks workgroups
        // Round 4, for the HBM-bound N = 320 linears with a residual (to_out / proj_out: A + residual + out = 86.5 MB against 4.6 GFLOP): tile
        // 34 splits the five N tiles of a row block over two workgroups, i.e. reads the A rows TWICE (115 MB in all).  Smaller row blocks
        // reach two workgroups per CU without an N split -- A, residual and out each cross the fabric once -- at the price of fewer
        // MFMAs per W fragment read (2 / 1 instead of 3: irrelevant at 0.05 FLOP per byte).
        case 35: return launch_rg<4, 1, 2, 64, 5>(a, st);    // four waves of 32 rows: 128-row blocks (352 workgroups at M = 45056), all N tiles
        case 36: return launch_rg<4, 1, 1, 64, 5>(a, st);    // four waves of 16 rows: 64-row blocks (704 workgroups), all N tiles
#ifdef PCDM_DEV_ROWGEMM_VARIANTS
        case 40: return launch_rg<4, 2, 3, 128, 4>(a, st);   // as 31 with a 4-stage ring
        case 41: return launch_rg<4, 2, 3, 128, 3>(a, st);   // as 31 with a 3-stage ring
        case 42: return launch_rg<4, 1, 3, 64, 3, 512>(a, st);   // as 34 with a 3-stage ring
        case 43: return launch_rg<4, 1, 3, 64, 4, 512>(a, st);   // as 34 with a 4-stage ring
        case 44: return launch_rg<4, 1, 3, 64, 5, 768>(a, st);   // as 34 with the N tiles split three ways at M = 45056
#endif
        default: return -1;
    }
}
