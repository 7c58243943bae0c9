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
//  * one workgroup = 8 waves as WGM x WGN; a wave owns FMW*16 = 48 rows for the whole launch and keeps them as MFMA operand fragments
//    in registers (16x16x32: 10 k-steps x 3 row blocks x 4 VGPRs = 120); the A rows are read from HBM ONCE, straight into registers;
//  * LayerNorm (optional, template LNF) is applied to those registers right after the load -- a row is spread over 4 lanes (80
//    elements each): local sums, two xor-shuffles, exact two-pass variance -- and the normalised rows never exist in memory;
//  * the workgroup then walks over ALL N tiles (BN = 64 / 128 columns): W tiles stream HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds, as
//    gemm.hip) through ONE ring of NSTG stages of 64 k that runs across N-tile boundaries: the next tile's first stages are in flight
//    while the current tile's epilogue runs, 5-7 stages (40-96 KiB) outstanding per CU; only B fragments are read from LDS
//    (FN ds_read_b128 per FN*FMW MFMAs);
//  * M = 45056 gives 235 workgroups of 192 rows (92 % of the CUs, one round); the WGM = 2 shape (96 rows) does the same for the
//    half-batch to_q of the cross-attention (M = 22528);
//  * epilogues: bias (+ residual) store, GEGLU ([32 h | 32 gate] packed rows, as gemm.hip), q|k store + V^T transposed store -- all
//    through a wave-private fp32 LDS tile so that every global access is 16 B per lane along the contiguous axis.
// vmcnt bookkeeping: vector-memory operations of a wave retire from its counter in issue order, loads and stores alike (gfx9 has one
// counter for both and hipcc's own waits rely on that order).  The ring's counted waits therefore count, besides the younger ring
// stages, the epilogue STORES issued since the awaited stage (a fixed number per N tile); loads the epilogue issues (bias, residual)
// are not counted: an uncounted operation can only make a wait longer than necessary, never shorter.  The first version counted ring
// stages only: every wait that followed an epilogue then also waited for that epilogue's store acknowledgements (~1.5 us per N tile).
#include "gemm_args.h"

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

// s_waitcnt vmcnt(n) + lgkmcnt(0), then s_barrier; n is wave-uniform at run time (a scalar switch over immediates)
__device__ __forceinline__ void wait_stage(int n) {
#ifdef PCDM_EMU
    (void)n;
    __syncthreads();
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
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ void wait_lds_rg() {   // s_waitcnt lgkmcnt(0), visible to the compiler's scoreboard
#ifndef PCDM_EMU
    __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));
#endif
}

// WGM x WGN waves; a wave owns FMW row blocks of 16 and, per N tile, BN / WGN columns; NSTG ring stages of [BN][64]; LNF: LayerNorm on A
template <int WGM, int WGN, int FMW, int BN, int NSTG, bool LNF>
__global__ __launch_bounds__(WGM* WGN * 64) void rowgemm_kernel(const GemmArgs p) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * FMW * 16;
    constexpr int WNC = BN / WGN;          // columns of an N tile per wave
    constexpr int FN = WNC / 16;           // B fragments per wave per k-step
    constexpr int DPW = BN / 8 / NW;       // LDS-DMA instructions per wave per stage (8 rows of 128 B each)
    constexpr int EPW = WNC + 4;           // fp32 pitch of the wave-private epilogue tile (16 rows)
    static_assert(NW == 8 && BN % (8 * NW) == 0 && WNC % 16 == 0 && (WNC == 32 || WNC == 64), "shape");
    static_assert(NSTG >= 3 && (NSTG - 2) * DPW <= 14, "ring depth");
    PCDM_DYN_SMEM(smem);
    u16* Ws = (u16*)smem;                                  // [NSTG][BN][64]  (unpadded, 16-byte chunks XOR-swizzled by (row >> 1) & 7)
    float* eps_all = (float*)(Ws + NSTG * BN * 64);        // [NW][16][EPW]
    float* bias_s = eps_all + NW * 16 * EPW;               // [Npad]: the epilogue reads no global memory (a load behind a store would
                                                           // wait for that store's acknowledgement: vmcnt retires in issue order)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
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
    const int NT = p.Npad / BN;
    for (int i = t * 4; i < p.Npad; i += NW * 64 * 4) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(bias_s + i) = p.bias ? *(const f32x4*)(p.bias + i) : z4;
    }
    const bool skip = m0 - wm * (FMW * 16) + BM <= p.zero_rows;   // every row of the workgroup is declared zero: epilogue only
    const int Q = skip ? 0 : NT * kNKT;
    auto issue = [&](int q) {
        const int nt = q / kNKT, kt = q - nt * kNKT;
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
    if constexpr (LNF) {
        // LayerNorm over the 320 elements of each row (4 lanes x 80): mean, then the centred sum of squares (exact two-pass, fp32),
        // then (x - mean) * rstd * gamma + beta rounded to bf16 -- the values pcdm_layernorm would have written to memory
        float mean[FMW], rstd[FMW];
#pragma unroll
        for (int j = 0; j < FMW; ++j) {
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += bf2f(xa[j][ks][e]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            mean[j] = s * (1.0f / kK);
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) RG_KEEP_PACKED(xa[j][ks]);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = bf2f(xa[j][ks][e]) - mean[j];
                    q += d * d;
                }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            rstd[j] = 1.0f / sqrtf(q * (1.0f / kK) + p.ln_eps);
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) RG_KEEP_PACKED(xa[j][ks]);
        }
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            const int k = ks * 32 + 8 * lq;
            const f32x4 g0 = *(const f32x4*)(p.ln_gamma + k), g1 = *(const f32x4*)(p.ln_gamma + k + 4);
            const f32x4 c0 = *(const f32x4*)(p.ln_beta + k), c1 = *(const f32x4*)(p.ln_beta + k + 4);
#pragma unroll
            for (int j = 0; j < FMW; ++j) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = e < 4 ? g0[e] : g1[e - 4], c = e < 4 ? c0[e] : c1[e - 4];
                    y[e] = (bf2f(xa[j][ks][e]) - mean[j]) * rstd[j] * g + c;
                }
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2bf(y[2 * e], y[2 * e + 1]);
                xa[j][ks] = __builtin_bit_cast(u16x8, o);
                RG_KEEP_PACKED(xa[j][ks]);
            }
        }
    }

    // ---- epilogue operands
    const bool geglu = p.epilogue == PCDM_EPI_GEGLU;
    const int vt0 = p.epilogue == PCDM_EPI_SPLIT_VT ? p.vt_col0 : 0x7fffffff;   // columns >= vt0 go to out2 transposed
    const int ncols_out = p.epilogue == PCDM_EPI_SPLIT_VT ? p.vt_col0 : (geglu ? p.N : p.N);
    const bool has_res = p.residual != nullptr && !geglu;
    const BufRsrc rs_o = make_buf_rsrc(p.out, (uint32_t)((((int64_t)p.M - 1) * p.ldo + ncols_out) * 2));
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
    auto epilogue = [&](int nt) {
        const int n0w = nt * BN + wn * WNC;                 // first (packed) column of this wave's tile
        if (geglu) {
            if constexpr (WNC == 64) {
                // packed rows alternate [32 h | 32 gate]: fragments 0, 1 = h, 2, 3 = gate of output channels n0w / 2 .. + 31
                f32x4 bh[2], bg[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    bh[i] = *(const f32x4*)(bias_s + n0w + i * 16 + 4 * lq);
                    bg[i] = *(const f32x4*)(bias_s + n0w + 32 + i * 16 + 4 * lq);
                }
                const bool swiglu = p.act == PCDM_ACT_SILU;   // (wave-uniform: one branch per row block, not one per element)
                const int no = n0w / 2 + (lane & 3) * 8;    // read-back: 4 lanes per row, 16 rows per instruction
                const int rr = lane >> 2;
#pragma unroll
                for (int j = 0; j < FMW; ++j) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x4 v;
                        if (swiglu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][e] + bh[i][e]) * silu_f(acc[i + 2][j][e] + bg[i][e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][e] + bh[i][e]) * gelu_erf_f(acc[i + 2][j][e] + bg[i][e]);
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
        }
        if (n0w >= vt0) {
            // V^T: a lane takes one channel and 8 consecutive tokens; out2[b, channel, token]
#pragma unroll
            for (int j = 0; j < FMW; ++j) {
                const int mb = m0 + j * 16;                 // 16 rows inside one batch entry (rows_per_batch % 16 == 0)
                const int b = mb / p.rows_per_batch, tok0 = mb - b * p.rows_per_batch;
#pragma unroll
                for (int i = 0; i < FN; ++i) *(f32x4*)(ep + lrow * EPW + i * 16 + 4 * lq) = acc[i][j];
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
                rv[j][it] = buf_load16(rs_r, nok ? (uint32_t)(((int64_t)m * p.ldr + n) * 2) : kOOB);
            }
#pragma unroll
        for (int j = 0; j < FMW; ++j) {
#pragma unroll
            for (int i = 0; i < FN; ++i) *(f32x4*)(ep + lrow * EPW + i * 16 + 4 * lq) = acc[i][j];
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
    };

    // stores every epilogue is guaranteed to issue per wave (masked lanes / rows beyond M still issue the instruction)
    const int n_ep_stores = geglu ? FMW : FMW * (WNC / 32);
    if (skip) {   // all rows zero: out = bias (+ residual)
        __syncthreads();   // (bias_s)
        for (int nt = 0; nt < NT; ++nt) epilogue(nt);
        return;
    }

    // ---- main loop: for every stage q: [wait: stage q landed, everybody done with stage q - 1] -> refill the freed slot with stage
    // q + NSTG - 1 -> 2 k-steps of MFMAs from registers (A) x LDS (W).  The N-tile loop is a run-time loop, its 5 K-tiles are unrolled
    // (the A fragments are indexed by compile-time k-step numbers: registers).
    const int frow_sw = (lrow >> 1) & 7;
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int kt = 0; kt < kNKT; ++kt) {
            const int q = nt * kNKT + kt;
            // operations this wave may leave in flight: the younger ring stages q + 1 .. min(q + NSTG - 2, Q - 1), and the stores of
            // the epilogues it ran since it issued stage q (in iteration lo = q - NSTG + 1, or in the prologue): an iteration i ends with
            // an epilogue iff i % kNKT == kNKT - 1, so there are q / kNKT - lo / kNKT of them in [lo, q - 1]
            const int younger = (Q - 1 - q) < (NSTG - 2) ? (Q - 1 - q) : (NSTG - 2);
            const int lo = q - (NSTG - 1) > 0 ? q - (NSTG - 1) : 0;
            wait_stage(younger * DPW + (q / kNKT - lo / kNKT) * n_ep_stores);
            if (q + NSTG - 1 < Q) issue(q + NSTG - 1);
            const u16* ws = Ws + (q % NSTG) * (BN * 64) + (wn * WNC + lrow) * 64;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                u16x8 wf[FN];
                const int co = ((k2 * 4 + lq) ^ frow_sw) * 8;
#pragma unroll
                for (int i = 0; i < FN; ++i) wf[i] = *(const u16x8*)(ws + i * 16 * 64 + co);
                wait_lds_rg();
                PCDM_SCHED_BARRIER();
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FMW; ++j) acc[i][j] = mfma_16x16x32(wf[i], xa[j][kt * 2 + k2], acc[i][j]);
                PCDM_SCHED_BARRIER();
            }
        }
        epilogue(nt);
        zero_acc();
    }
}

template <int WGM, int WGN, int FMW, int BN, int NSTG>
int launch_rg(const GemmArgs& a, hipStream_t st) {
    constexpr int BM = WGM * FMW * 16, WNC = BN / WGN, NW = WGM * WGN;
    constexpr int smem_fixed = NSTG * BN * 64 * (int)sizeof(u16) + NW * 16 * (WNC + 4) * (int)sizeof(float);
    if (a.Npad % BN || a.Npad > 4096) return -1;
    const int smem = smem_fixed + a.Npad * (int)sizeof(float);   // + the bias vector
    if (a.epilogue == PCDM_EPI_GEGLU && WNC != 64) return -1;
    if (a.epilogue == PCDM_EPI_SPLIT_VT && (a.vt_col0 % WNC || a.rows_per_batch % 16 || (a.ldo2 & 7) || a.M % 16)) return -1;
    const int grid = (a.M + BM - 1) / BM;
    if (a.ln_gamma) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)rowgemm_kernel<WGM, WGN, FMW, BN, NSTG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_fixed + 4096 * (int)sizeof(float));
            attr_done = true;
        }
        PCDM_LAUNCH(PCDM_KERNEL_NAME(rowgemm_kernel<WGM, WGN, FMW, BN, NSTG, true>), dim3(grid), dim3(NW * 64), smem, st, a);
    } else {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)rowgemm_kernel<WGM, WGN, FMW, BN, NSTG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_fixed + 4096 * (int)sizeof(float));
            attr_done = true;
        }
        PCDM_LAUNCH(PCDM_KERNEL_NAME(rowgemm_kernel<WGM, WGN, FMW, BN, NSTG, false>), dim3(grid), dim3(NW * 64), smem, st, a);
    }
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
    if (a.ln_gamma && (!a.ln_beta || a.zero_rows)) return -1;
    switch (tile) {
        case 31: return launch_rg<4, 2, 3, 128, 6>(a, st);   // 192 rows, N tiles of 128 (waves 48 x 64: GEGLU-capable), 96 + 34 KiB
        case 32: return launch_rg<4, 2, 3, 64, 8>(a, st);    // 192 rows, N tiles of 64 (waves 48 x 32), 64 + 18 KiB
        case 33: return launch_rg<2, 4, 3, 128, 6>(a, st);   // 96 rows, N tiles of 128 (waves 48 x 32): M = 22528 -> 235 workgroups
        case 34: return launch_rg<2, 4, 3, 256, 3>(a, st);   // 96 rows, N tiles of 256 (waves 48 x 64: GEGLU-capable), 96 + 34 KiB
        default: return -1;
    }
}
