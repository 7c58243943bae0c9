// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution with fused epilogues (SURVEY.md §2.1 K1-K3,K5-K7,K11).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),  fp32 accumulation on v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16.
//
// Design (gfx950):
//  * One workgroup = 8 waves (block tiles 192x320, 192x256, 256x256, 256x128, 256x64, 512x64, 128x128) or 4 / 2 waves (128x128, 128x64,
//    64x64, 256x64), BK = 64.  Each wave owns a 96x80 / 96x64 sub-tile as 16x16x32 fragments or a 128x64 .. 32x32 one as 32x32x16
//    fragments.  The MFMA is issued "swapped" -- D = Wfrag * Xfrag^T, rows = output channel n, cols = pixel/token m -- so that every
//    lane ends up with 4 CONSECUTIVE output channels of one row in each accumulator quad.
//  * A (activations, NHWC bf16) and W (packed [N][K], K contiguous) tiles go HBM/L2 -> LDS by LDS-DMA through buffer descriptors
//    (buffer_load_dwordx4 ... offen lds, 16 B per lane, no VGPR round trip) into a ring of STAGES LDS buffers; ONE s_barrier per
//    K-tile preceded by a COUNTED s_waitcnt vmcnt(N) so the younger tiles' DMAs stay in flight across it.  The DMA destination is
//    lane-linear, the per-lane SOURCE offset is free -- that is where the implicit-GEMM gather lives: 3x3 halo / zero padding (an
//    out-of-range offset returns zeros), stride 2, nearest-x2 upsample folding, the two-source skip concat and `zero_rows` are all
//    just per-lane source offsets.  The K loop itself ("rotated": barrier in front of the last k-step's MFMAs) is described at the loop.
//  * LDS tiles are unpadded [rows][64] bf16 with the 16-byte chunk index XOR-swizzled by (row>>1)&7,
//    applied on the DMA source offset and again on the fragment reads (cdna_hip_programming.md rule 21):
//    the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-byte slots of the 256-byte bank row.
//  * Epilogue: accumulators -> wave-private fp32 LDS tile -> 16-byte rows; bias / time-embedding row / residual / GEGLU in registers;
//    buffer stores (no predicates); V^T of the fused q|k|v projection written transposed.  Split-K partials go to an fp32 workspace.
//  * Workgroup ids are remapped so each XCD (private 4 MiB L2) owns a contiguous run of M-tiles and
//    walks all N-tiles of an A tile back to back (cdna_hip_programming.md T1, bijective form).
//  * What bounds it: with real operands the MFMA + LDS loop without any global loads sustains 1.1-1.5 PF/s (power).  The
//    L2 -> LDS fill path is NOT the limiter (profiles/r3_pmc_fill.json: TA busy 2-6 %, TCC busy 7-11 % on every UNet shape); the thin-K
//    launches are bounded by their prologue / epilogue around a five-tile K loop and by one barrier domain per CU (tools/gemm_anatomy.py).
#include "gemm_args.h"

#include <type_traits>



namespace {
using pcdm_gemm_detail::GemmArgs;
using pcdm_gemm_detail::apply_act;
using pcdm_gemm_detail::gate_act;
constexpr int BK = 64;

// 32 bytes of zeros: the source of every epilogue operand that is absent (no bias / row vector / residual) or out of range, so
// that the epilogue's loads are UNCONDITIONAL: a load inside `if (p.bias)` costs a branch plus an s_waitcnt vmcnt(0) of its own
// (hipcc never batches loads across such branches), i.e. one exposed L2 / HBM round trip per operand per 8 outputs
__device__ __attribute__((aligned(32))) const unsigned int g_zero32[8] = {0, 0, 0, 0, 0, 0, 0, 0};

// One pass of the LDS-staged epilogue: the wave's fp32 tile (32 rows x WCOLS channels, pitch EPW) is read back row-major, 16 B
// (8 channels) per lane; out = acc + bias + rowvec + residual as bf16 (activations take the generic path).  Written to be LEAN --
// tools/gemm_anatomy.py (s_memtime stamps) showed the first version spending 15.7k-23.6k cycles per 96x80 wave tile, more than the
// five K-tiles of a K = 320 main loop, on: a branch + s_waitcnt vmcnt(0) around every optional operand load, a three-way activation
// switch per element, 64-bit address arithmetic and row predicates.  Here:
//  * residual / row vector / output go through buffer descriptors sized to the tensors: rows >= M and masked lanes (offset bit 31)
//    are dropped / read as zero by the bounds check (which applies to the per-lane offset, so the row term lives there);
//  * HAS_RES / HAS_RV are compile-time (the caller branches once, wave-uniformly); every residual load of the pass is issued first;
//  * 32-bit offsets, one add per store instruction.
// geometry of a pass: lane -> (row within the store instruction, first of its 8 channels)
template <int WCOLS>
struct PassGeom {
    static constexpr int LPR = WCOLS / 8;                 // lanes per output row
    static constexpr int RPI = 64 / LPR;                  // rows per store instruction (WCOLS = 48: 10, the last 4 lanes idle)
    static constexpr int NIT = (32 + RPI - 1) / RPI;      // store instructions per pass (<= 4)
    static constexpr bool TAIL = NIT * RPI > 32;          // the last instruction covers rows beyond the pass (WCOLS = 48 only)
};

// residual rows of one pass -> registers (issued two passes ahead of their use: they come from HBM; a descriptor of size 0 --
// no residual -- returns zeros without touching memory, so the loads are unconditional)
template <int WCOLS>
__device__ __forceinline__ void lean_res_load(const GemmArgs& p, int lane, int mrow0, int ncol0, BufRsrc rs_r, u32x4 (&rv)[4]) {
    typedef PassGeom<WCOLS> G;
    constexpr uint32_t kOOB = 0x80000000u;
    const int rl = lane / G::LPR, c8 = (lane - rl * G::LPR) * 8;
    const int n = ncol0 + c8;
    const bool lane_ok = rl < G::RPI && n < p.N;
    const uint32_t vr0 = lane_ok ? (uint32_t)(((mrow0 + rl) * (int)p.ldr + n) * 2) : kOOB;
    const uint32_t sr = (uint32_t)(G::RPI * (int)p.ldr * 2);
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) rv[it] = buf_load16(rs_r, (G::TAIL && it * G::RPI + rl >= 32) ? kOOB : vr0 + it * sr);
}

// RV: 0 = no row vector, 1 = row vector from the wave's LDS slice (rows b_lo and b_lo + 1 of it, staged before the first store), 2 = from
// global memory per store instruction (wave tiles that span more than two batch entries: the 8x11 level)
template <int WCOLS, int EPW, int RV>
__device__ __forceinline__ void lean_pass(const GemmArgs& p, const float* ep, int lane, int mrow0, int ncol0, f32x4 b0, f32x4 b1,
                                          BufRsrc rs_o, BufRsrc rs_v, const u32x4 (&rv)[4], const float* rvec_w, int rv_pitch,
                                          int rv_col0, int rv_split_row) {
    typedef PassGeom<WCOLS> G;
    constexpr int LPR = G::LPR, RPI = G::RPI, NIT = G::NIT;
    constexpr bool TAIL = G::TAIL;
    constexpr uint32_t kOOB = 0x80000000u;
    const int rl = lane / LPR, c8 = (lane - rl * LPR) * 8;
    const int n = ncol0 + c8;
    const bool lane_ok = rl < RPI && n < p.N;
    const uint32_t vo0 = lane_ok ? (uint32_t)(((mrow0 + rl) * (int)p.ldo + n) * 2) : kOOB;
    const uint32_t so = (uint32_t)(RPI * (int)p.ldo * 2);
    f32x4 t0[2], t1[2];
    if constexpr (RV == 1) {   // the two candidate rows of the time-embedding projection, from LDS: no vector-memory load behind a store
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            t0[h] = *(const f32x4*)(rvec_w + h * rv_pitch + rv_col0 + c8);
            t1[h] = *(const f32x4*)(rvec_w + h * rv_pitch + rv_col0 + c8 + 4);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int r = it * RPI + rl;
        const int rr = (TAIL && r >= 32) ? 0 : r;
        const f32x4 v0 = *(const f32x4*)(ep + rr * EPW + c8), v1 = *(const f32x4*)(ep + rr * EPW + c8 + 4);
        f32x4 a0 = v0 + b0, a1 = v1 + b1;
        if constexpr (RV == 1) {
            const bool hi = mrow0 + r >= rv_split_row;   // first row of batch entry b_lo + 1
            a0 += hi ? t0[1] : t0[0];
            a1 += hi ? t1[1] : t1[0];
        }
        if constexpr (RV == 2) {   // rows of one instruction span at most two batch entries (rows_per_batch >= 32 on this path)
            const int mb = mrow0 + it * RPI;                        // wave-uniform
            const int b_lo = mb / p.rows_per_batch;
            const int bidx = b_lo + ((mb + rl) >= (b_lo + 1) * p.rows_per_batch ? 1 : 0);
            const uint32_t vv = lane_ok ? (uint32_t)((bidx * p.ldrv + n) * 4) : kOOB;
            a0 += __builtin_bit_cast(f32x4, buf_load16(rs_v, vv));
            a1 += __builtin_bit_cast(f32x4, buf_load16(rs_v, vv + 16));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // residual (zeros when there is none)
            a0[e] += __builtin_bit_cast(float, rv[it][e >> 1] << (e & 1 ? 0 : 16) & 0xffff0000u);
            a1[e] += __builtin_bit_cast(float, rv[it][2 + (e >> 1)] << (e & 1 ? 0 : 16) & 0xffff0000u);
        }
        u32x4 o;
        o[0] = pack2bf(a0[0], a0[1]);
        o[1] = pack2bf(a0[2], a0[3]);
        o[2] = pack2bf(a1[0], a1[1]);
        o[3] = pack2bf(a1[2], a1[3]);
        buf_store16(rs_o, (TAIL && r >= 32) ? kOOB : vo0 + it * so, o);
    }
}

// V^T pass (PCDM_EPI_SPLIT_VT, columns >= vt_col0): the staged 32 tokens x WCOLS channels are read back COLUMN-wise -- a lane takes one
// channel and 8 consecutive tokens -- and stored as 16 bytes along the token axis of out2[b, channel, token]: 64 contiguous bytes
// per channel per pass instead of 2-byte scalar stores.  The 32 rows of a pass lie inside one batch entry (rows_per_batch % 32 == 0).
template <int WCOLS, int EPW>
__device__ __forceinline__ void vt_pass(const GemmArgs& p, const float* ep, int lane, int mrow0, int ncol0, BufRsrc rs_vt,
                                        const float* bias_c /* the wave's LDS bias slice at column ncol0 */) {
    constexpr uint32_t kOOB = 0x80000000u;
    const int b = mrow0 / p.rows_per_batch, tok0 = mrow0 - b * p.rows_per_batch;   // wave-uniform
    const int cl = lane & 15, tg = lane >> 4;
    const int cv = p.N - p.vt_col0;
#pragma unroll
    for (int ii = 0; ii < WCOLS / 16; ++ii) {
        const int c = ii * 16 + cl, n = ncol0 + c;
        const bool ok = n < p.N;
        const float bias = bias_c[c];
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ep[(8 * tg + i) * EPW + c] + bias;
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pack2bf(v[2 * i], v[2 * i + 1]);
        const uint32_t vo = ok ? (uint32_t)((((int64_t)b * cv + (n - p.vt_col0)) * p.ldo2 + tok0 + 8 * tg) * 2) : kOOB;
        buf_store16(rs_vt, vo, o);
    }
}

template <int N>
__device__ __forceinline__ void wait_vm_then_barrier() {
    // counted wait (N LDS-DMA instructions of younger tiles may stay in flight) + raw s_barrier in ONE asm
    // statement: __syncthreads() would drain vmcnt to 0 (cdna_hip_programming.md "Pipelining across barriers")
#ifdef PCDM_EMU
    __syncthreads();
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

template <int N>
__device__ __forceinline__ void wait_vm_lds_then_barrier() {   // as above, and this wave's ds_reads are complete too
#ifdef PCDM_EMU
    __syncthreads();
#else
    // builtins, not inline asm: the compiler's own waitcnt insertion must SEE that the LDS counter was drained here -- behind an opaque
    // asm it assumed the previous k-step's fragment reads were still pending and put an s_waitcnt lgkmcnt(0) between every block of
    // ds_reads and the MFMAs that were meant to cover them (simm16: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ void wait_lds() {   // s_waitcnt lgkmcnt(0) (vmcnt / expcnt untouched), visible to the compiler's scoreboard
#ifndef PCDM_EMU
    __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));
#endif
}

template <int N>
__device__ __forceinline__ void wait_vm() {
#ifndef PCDM_EMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void wait_lds_then_barrier() {  // this wave's ds_reads are complete, then raw s_barrier
#ifdef PCDM_EMU
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// STAG (8-wave tiles only): the two waves that share a SIMD (w and w+4) run half a K-tile out of phase --
// while one is in its MFMA segment (16 back-to-back MFMAs on fragments held in registers) the other is in its
// memory segment (LDS-DMA issue for tile t+D, 16 ds_read_b128 of tile t), two raw barriers per K-tile
// (MI355X_MICROARCH.md "Two waves per SIMD").  Without it every wave alternates memory and matrix phases in
// lockstep and the two pipes are used one after the other (profiles/r1_gemm_ablation.txt: full ~ noload + nomfma).
// F = MFMA fragment edge: 32 (v_mfma_f32_32x32x16_bf16, 16 accumulator registers per fragment) or 16 (v_mfma_f32_16x16x32_bf16, 4):
// the same FLOP rate and the same LDS bytes per FLOP for a given wave tile; F = 16 allows wave tiles that are multiples of 16
// (96x80: the 192x320 / 96x320 block tiles, whose counts divide the 256 CUs for M = 45056 / 11264 / 2816).
// KB = K-tile depth.  KB = 32 (F = 16, four stages): the same LDS bytes as two 64-deep stages, but THREE tiles of 32 in flight behind
// the one being multiplied instead of one tile of 64 -- 96 KiB instead of 64 KiB of loads outstanding per CU.  The KB = 64 loop of the
// 192x320 tile takes ~2.8 k cycles per K-tile whatever the problem (tools/gemm_anatomy.py), against 1.9 k cycles of MFMA issue: one
// memory latency per K-tile, i.e. the ring is too shallow, not the matrix pipe too slow.  Loader ROLES: a DMA instruction covers
// 16 rows of 64 bytes, so the (BM + BN) / 16 instructions of a tile are dealt out whole-operand -- waves [0, NWA) stage A, the rest B.
template <int BM, int BN, int WGM, int WGN, int STAGES, bool CONV, bool STAG = false, int F = 32, int KB = 64>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_kernel(const GemmArgs p) {
    constexpr int BK = KB;                 // (shadows the file-scope default of 64)
    constexpr int NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / F, FN = WN / F;
    constexpr int KS = 512 / F;            // k per MFMA: 16 (32x32) or 32 (16x16)
    constexpr int NKS = BK / KS;           // MFMA k-steps per K-tile
    constexpr int CPK = KS / 8;            // 16-byte chunks per fragment row per k-step (= lanes / F)
    constexpr int NQ = F == 32 ? 4 : 1;    // accumulator quads (4 consecutive channels of one pixel) per lane per fragment
    constexpr int LF = F == 32 ? 5 : 4;    // log2(F)
    static_assert((F == 32 || F == 16) && (!STAG || F == 32), "fragment shape");
    static_assert(KB == 64 || (KB == 32 && F == 16 && STAGES == 4 && !STAG), "K-tile depth");
    typedef typename std::conditional<F == 32, f32x16, f32x4>::type acc_t;
    constexpr bool ROLES = KB == 32;                   // loader roles (see above)
    constexpr int LPR = BK / 8;                        // lanes (16-byte chunks) per staged row
    constexpr int RPI = 64 / LPR;                      // rows per LDS-DMA wave-instruction: 8 (128-byte rows) or 16 (64-byte rows)
    constexpr int PW = ROLES ? (BM + BN) / (RPI * NW) : 0;
    constexpr int NWA = ROLES ? BM / (RPI * PW) : NW;  // waves staging A
    // LDS-DMA instructions per wave per K-tile, A rows / B rows
    constexpr int AI = ROLES ? PW : BM / 8 / NW, BI = ROLES ? PW : BN / 8 / NW;
    constexpr int PWT = ROLES ? PW : AI + BI;          // ... in total (every wave issues the same number: counted vmcnt waits)
    constexpr int D = STAGES - 1;                      // prefetch distance (tiles in flight)
    static_assert(ROLES || (BM % (8 * NW) == 0 && BN % (8 * NW) == 0), "tile shape");
    static_assert(!ROLES || ((BM + BN) % (RPI * NW) == 0 && BM % (RPI * PW) == 0 && BN % (RPI * PW) == 0), "tile shape (roles)");
    static_assert(WM % 32 == 0 && WN % F == 0, "tile shape");
    PCDM_DYN_SMEM(smem);
    u16* As = (u16*)smem;                    // [STAGES][BM][BK]   (unpadded, XOR-swizzled 16-byte chunks)
    u16* Bs = As + STAGES * BM * BK;         // [STAGES][BN][BK]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // provably wave-uniform -> LDS bases / branches in SGPRs
    const int wm = wave / WGN, wn = wave - wm * WGN;
#ifndef PCDM_EMU
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
#define PCDM_STAMP(i) do { if (p.debug & 4) stamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PCDM_STAMP(i) ((void)0)
#endif
    PCDM_STAMP(0);

    // XCD-aware bijective remap of the linear workgroup id
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ntiles = p.tiles_m * p.tiles_n;
    const int ksplit = wg / ntiles, tile_id = wg - ksplit * ntiles;  // neighbours = same K slice, adjacent N tiles
    const int tile_m = tile_id / p.tiles_n, tile_n = tile_id - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- LDS-DMA staging (buffer_load_dwordx4 ... lds).  Wave-instruction j of wave w fills tile rows
    // (w*AI+j)*8 .. +7: lane i -> LDS slot i = (row i>>3, position i&7); position q of row r holds global chunk
    // q ^ ((r>>1)&7) (swizzle on the SOURCE offset, linear destination; the same XOR is applied on the fragment
    // reads).  Addressing = one buffer descriptor per operand (SGPRs) + a per-lane 32-bit byte offset that is
    // CONSTANT over the K loop + a wave-uniform SGPR offset that advances with the K-tile: no per-lane address
    // arithmetic in the steady state.  Out-of-range offsets return zeros (hardware bounds check), which is how
    // the implicit-GEMM zero padding (3x3 halo) is produced.
    const int srow = lane / LPR, spos = lane % LPR;
    // chunk swizzle of tile row r: 128-byte rows (r >> 1) & 7; 64-byte rows (-(r >> 2)) & 3 -- with either, the 16 lanes that one
    // ds_read_b128 cycle serves ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md "LDS") hit 16 different 16-byte bank groups
    auto swz = [](int r) { return KB == 64 ? (r >> 1) & 7 : (0 - (r >> 2)) & 3; };
    const bool doA = !ROLES || wave < NWA, doB = !ROLES || wave >= NWA;   // wave-uniform
    const int wa = ROLES ? (wave < NWA ? wave : 0) : wave, wb = ROLES ? (wave >= NWA ? wave - NWA : 0) : wave;
    constexpr uint32_t kOOB = 0x80000000u;
    uint32_t a_off[AI], a_off2[AI];   // byte offsets: linear: row*lda (+chunk) in a / a2; conv: centre tap pixel
    int a_mask[AI];                   // conv: bit t set <=> tap t (= ky*3+kx) of this row is inside the image
    int a_b[AI], a_y[AI], a_x[AI];    // conv + upsample only: coordinates for the per-tile gather
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int rl = (wa * AI + j) * RPI + srow;
        int m = m0 + rl;
        const uint32_t ck = (uint32_t)(spos ^ swz(rl)) * 16u;   // bytes
        const bool mvalid = m < p.M;
        if (!mvalid) m = p.M - 1;    // rows >= M are never stored: any in-range data will do
        a_off2[j] = 0; a_mask[j] = 0; a_b[j] = a_y[j] = a_x[j] = 0;
        if (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / p.Wo, x = rem - y * p.Wo;
            a_b[j] = b; a_y[j] = y; a_x[j] = x;
            // offset of tap (0,0) relative to a descriptor base shifted back by (Wi+1) pixels (see rs_a below)
            a_off[j] = (uint32_t)((((int64_t)b * p.Hi + y * p.stride) * p.Wi + x * p.stride) * p.cin * 2) + ck;
            const int Hv_ = p.upsample ? p.Ho : p.Hi, Wv_ = p.upsample ? p.Wo : p.Wi;   // (virtual) input extent
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int iy = y * p.stride + tp / 3 - p.pad, ix = x * p.stride + tp % 3 - p.pad;
                if (iy >= 0 && iy < Hv_ && ix >= 0 && ix < Wv_) a_mask[j] |= 1 << tp;
            }
        } else {
            // rows the caller declared all-zero are "out of range": the bounds check delivers zeros, nothing is fetched
            a_off[j] = m < p.zero_rows ? kOOB : (uint32_t)((int64_t)m * p.lda * 2) + ck;
            a_off2[j] = m < p.zero_rows ? kOOB : (uint32_t)((int64_t)m * p.lda2 * 2) + ck;
        }
    }
    uint32_t b_off[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int rl = (wb * BI + j) * RPI + srow;
        b_off[j] = (uint32_t)((int64_t)(n0 + rl) * p.ldw * 2) + (uint32_t)(spos ^ swz(rl)) * 16u;
    }
    // conv: descriptor base = a - (Wi+1) pixels, so that tap (ky,kx) is the NON-NEGATIVE uniform offset
    // (ky*Wi + kx)*cin*2; the bytes in front of the tensor are never touched (those taps are masked)
    const BufRsrc rs_a = make_buf_rsrc(CONV ? (const char*)p.a - (int64_t)p.pad * (p.Wi + 1) * p.cin * 2 : (const char*)p.a);
    const BufRsrc rs_a2 = make_buf_rsrc(p.a2 ? (const void*)p.a2 : (const void*)p.a);
    const BufRsrc rs_w = make_buf_rsrc(p.w);

    int kt0 = 0;  // first K-tile of this workgroup's K slice (set below)
    auto issue_tile = [&](int kt, int buf) {
        const int k0 = (kt0 + kt) * BK;
        u16* as = As + buf * BM * BK + (wa * AI) * RPI * BK;
        u16* bs = Bs + buf * BN * BK + (wb * BI) * RPI * BK;
        if (!doA) {
        } else if (CONV) {
            const int tap = k0 / p.cin, c0 = k0 - tap * p.cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            if (!p.upsample) {
                const uint32_t soff = (uint32_t)(((ky * p.Wi + kx) * p.cin + c0) * 2);
#pragma unroll
                for (int j = 0; j < AI; ++j)
                    buf_glds16(rs_a, ((a_mask[j] >> tap) & 1) ? a_off[j] : kOOB, soff, as + j * RPI * BK);
            } else {   // nearest upsample folded in (F.interpolate(mode="nearest") to Ho x Wo, then the conv): source pixel
                       // floor(v * Hi / Ho) -- v >> 1 for the usual x2 -- not affine in the tap
                const bool x2 = p.Ho == 2 * p.Hi && p.Wo == 2 * p.Wi;
#pragma unroll
                for (int j = 0; j < AI; ++j) {
                    const int vy = a_y[j] + ky - 1, vx = a_x[j] + kx - 1;
                    const int iy = x2 ? vy >> 1 : (vy > 0 ? vy * p.Hi / p.Ho : 0), ix = x2 ? vx >> 1 : (vx > 0 ? vx * p.Wi / p.Wo : 0);
                    const uint32_t ck = (uint32_t)(spos ^ swz((wa * AI + j) * RPI + srow)) * 16u;
                    const uint32_t off = (uint32_t)((((int64_t)a_b[j] * p.Hi + iy + 1) * p.Wi + ix + 1) * p.cin * 2) + ck;
                    buf_glds16(rs_a, ((a_mask[j] >> tap) & 1) ? off : kOOB, (uint32_t)(c0 * 2), as + j * RPI * BK);
                }
            }
        } else {
            const bool first = k0 < p.c1;
            const uint32_t soff = (uint32_t)((first ? k0 : k0 - p.c1) * 2);
            if (first) {
#pragma unroll
                for (int j = 0; j < AI; ++j) buf_glds16(rs_a, a_off[j], soff, as + j * RPI * BK);
            } else {
#pragma unroll
                for (int j = 0; j < AI; ++j) buf_glds16(rs_a2, a_off2[j], soff, as + j * RPI * BK);
            }
        }
        if (doB) {
#pragma unroll
            for (int j = 0; j < BI; ++j) buf_glds16(rs_w, b_off[j], (uint32_t)(k0 * 2), bs + j * RPI * BK);
        }
    };

    acc_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 4 * NQ; ++r) acc[i][j][r] = 0.f;

    // Wave tiles whose residual rows fit 32 VGPRs (64x32 / 32x64 / 64x64: the tiles the thin-K linears with a residual run on) request
    // them HERE, in front of the K loop: they do not depend on it, and loaded in the epilogue they are a full HBM round trip in front
    // of its first store (tools/gemm_anatomy.py: 16 k cycles of epilogue with a residual against 8 k without, on a 14 k-cycle K loop).
    constexpr int E_CG = FN < 64 / F ? FN : 64 / F;                          // fragment columns per epilogue pass (= CG below)
    constexpr int E_NPASS = ((FN + E_CG - 1) / E_CG) * (WM / 32);            // epilogue passes of the wave tile
    constexpr bool EARLY_RES = !CONV && E_NPASS * 16 <= 32;
    u32x4 rv_early[EARLY_RES ? E_NPASS : 1][4];
    bool early_res = false;
    if constexpr (EARLY_RES) {
        early_res = p.residual != nullptr && p.split_k <= 1 && p.epilogue == PCDM_EPI_STORE && p.act == 0 && (p.N & 7) == 0 && (p.ldo & 7) == 0 &&
                    (p.ldr & 7) == 0 && p.res_mod >= p.M && (!p.rowvec || p.rows_per_batch >= 32);
        if (early_res) {
            const BufRsrc rs_r0 = make_buf_rsrc(p.residual, (uint32_t)((((int64_t)p.M - 1) * p.ldr + p.N) * 2));
#pragma unroll
            for (int q = 0; q < E_NPASS; ++q) {
                constexpr int NJB_ = WM / 32;
                const int i0 = (q / NJB_) * E_CG;
                const int wc = ((FN - i0) < E_CG ? (FN - i0) : E_CG) * F;
                const int mrow0 = m0 + wm * WM + (q % NJB_) * 32, nc = n0 + wn * WN + i0 * F;
                if (wc == 64) lean_res_load<64>(p, lane, mrow0, nc, rs_r0, rv_early[q]);
                else if (wc == 32) lean_res_load<32>(p, lane, mrow0, nc, rs_r0, rv_early[q]);
                else if (wc == 16) lean_res_load<16>(p, lane, mrow0, nc, rs_r0, rv_early[q]);
                else lean_res_load<48>(p, lane, mrow0, nc, rs_r0, rv_early[q]);
            }
        }
    }

    const int nkt_all = p.K / BK;
    kt0 = (int)((int64_t)ksplit * nkt_all / p.split_k);
    int nkt = (int)((int64_t)(ksplit + 1) * nkt_all / p.split_k) - kt0;  // this workgroup's K-tiles
    if (!CONV && m0 + BM <= p.zero_rows) nkt = 0;   // the whole A tile is declared zero: epilogue only (bias + residual)
    // fragment reads: lane -> row (lane % F) of the fragment, 16-byte chunk (lane / F) of the k-step; (row>>1)&7 == (lane>>1)&7
    // because fragment rows are F-aligned (F = 16: ((lane & 15) >> 1) == (lane >> 1) & 7)
    const int frow = lane & (F - 1), fsw = swz(lane & (F - 1)), fhalf = lane >> LF;
    if constexpr (STAG) {
        static_assert(NW == 8 && D == 2, "staggered schedule: 8 waves, 3 stages");
        constexpr int PW = AI + BI;  // LDS-DMA instructions per wave per K-tile
        const bool grpB = wave >= NW / 2;
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (s < nkt) issue_tile(s, s);
        if (nkt > 1) wait_vm_then_barrier<PW>(); else wait_vm_then_barrier<0>();   // tile 0 landed for everyone
        if (grpB) wait_vm_then_barrier<PW>();   // group B lags one phase (vmcnt value irrelevant: already satisfied)
        int cur = 0, nxt = D % STAGES;
        for (int kt = 0; kt < nkt; ++kt) {
            // ---- memory segment: DMA for tile kt+D, all fragments of tile kt -> registers
            if (kt + D < nkt) issue_tile(kt + D, nxt);
            const u16* as = As + cur * BM * BK + (wm * WM + frow) * BK;
            const u16* bs = Bs + cur * BN * BK + (wn * WN + frow) * BK;
            u16x8 xf[BK / 16][FM], wf[BK / 16][FN];
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int co = ((ks * 2 + fhalf) ^ fsw) * 8;
#pragma unroll
                for (int j = 0; j < FM; ++j) xf[ks][j] = *(const u16x8*)(as + j * 32 * BK + co);
#pragma unroll
                for (int i = 0; i < FN; ++i) wf[ks][i] = *(const u16x8*)(bs + i * 32 * BK + co);
            }
            // tile kt+1 must have landed before the odd->even barrier (B: end of its memory segment,
            // A: end of its MFMA segment); one younger tile (kt+2) may stay in flight
            const bool more = kt + 2 < nkt;
            if (grpB) { if (more) wait_vm<PW>(); else wait_vm<0>(); }
            wait_lds_then_barrier();
            // ---- MFMA segment
            PCDM_SETPRIO(1);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) acc[i][j] = mfma_32x32x16(wf[ks][i], xf[ks][j], acc[i][j]);
            PCDM_SETPRIO(0);
            if (!grpB) { if (more) wait_vm<PW>(); else wait_vm<0>(); }
            wait_lds_then_barrier();
            cur = cur + 1 == STAGES ? 0 : cur + 1;
            nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
        }
        if (!grpB) wait_lds_then_barrier();   // matches group B's leading barrier
    } else if constexpr (KB == 32) {
        // four stages of 32: at the top of step kt the fragments of tile kt are in registers (read during step kt - 1), tile kt + 1
        // has landed, tiles kt + 2 and kt + 3 are in flight; the barrier frees the stage of tile kt for tile kt + 4, whose DMA is
        // issued first, then the fragment reads of tile kt + 1 (second register buffer), then the 30 MFMAs of tile kt cover both.
#pragma unroll
        for (int s = 0; s < STAGES; ++s)
            if (s < nkt) issue_tile(s, s);
        PCDM_STAMP(1);
        u16x8 xf[2][FM], wf[2][FN];
        const int co = (fhalf ^ fsw) * 8;
        auto load_frags = [&](int stage, auto bsel) {
            constexpr int b = decltype(bsel)::value;
            const u16* as = As + stage * BM * BK + (wm * WM + frow) * BK + co;
            const u16* bs = Bs + stage * BN * BK + (wn * WN + frow) * BK + co;
#pragma unroll
            for (int j = 0; j < FM; ++j) xf[b][j] = *(const u16x8*)(as + j * F * BK);
#pragma unroll
            for (int i = 0; i < FN; ++i) wf[b][i] = *(const u16x8*)(bs + i * F * BK);
        };
        auto wait_tiles = [&](int younger) {   // all but the `younger` most recent tiles of this wave have landed, then barrier
            if (younger >= 3) wait_vm_lds_then_barrier<3 * PWT>();
            else if (younger == 2) wait_vm_lds_then_barrier<2 * PWT>();
            else if (younger == 1) wait_vm_lds_then_barrier<PWT>();
            else wait_vm_lds_then_barrier<0>();
        };
        if (nkt > 0) {
            wait_tiles(nkt - 1 < 3 ? nkt - 1 : 3);
            PCDM_STAMP(2);
            load_frags(0, std::integral_constant<int, 0>());
        }
        auto step = [&](int kt, auto bsel) {
            constexpr int b = decltype(bsel)::value;
            const int left = nkt - 2 - kt;       // tiles issued beyond kt + 1
            wait_tiles(left < 0 ? 0 : (left < 2 ? left : 2));
            if (kt + STAGES < nkt && !(p.debug & 1)) issue_tile(kt + STAGES, kt & 3);
            if (kt + 1 < nkt) load_frags((kt + 1) & 3, std::integral_constant<int, b ^ 1>());
            PCDM_SCHED_BARRIER();
            if (!(p.debug & 2)) {
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) acc[i][j] = mfma_16x16x32(wf[b][i], xf[b][j], acc[i][j]);
            }
            PCDM_SCHED_BARRIER();
        };
        for (int kt = 0; kt < nkt; kt += 2) {
            step(kt, std::integral_constant<int, 0>());
            if (kt + 1 < nkt) step(kt + 1, std::integral_constant<int, 1>());
        }
    } else {
        // Rotated schedule: the workgroup barrier of a K-tile sits BEFORE the MFMAs of its last k-step, whose fragments are already in
        // registers -- the matrix pipe has work the moment the barrier opens, and the fragment reads of the next tile's first k-step
        // (issued right behind the barrier) return underneath it.  (With the barrier at the top of the tile every wave of the CU waited
        // out a full LDS round trip per K-tile with the matrix pipe idle: the loop without any global loads ran at 1.15 PF/s.)
        // At that barrier every wave has finished reading the stage of tile kt (all its k-steps are in registers or consumed), so the
        // stage is refilled with tile kt + STAGES straight away: STAGES - 1 tiles in flight behind the one being multiplied.
#pragma unroll
        for (int s = 0; s < STAGES; ++s)
            if (s < nkt) issue_tile(s, s);
        PCDM_STAMP(1);
        u16x8 xf[2][FM], wf[2][FN];
        auto load_frags = [&](int stage, int ks, auto bsel) {
            constexpr int b = decltype(bsel)::value;
            const int co = ((ks * CPK + fhalf) ^ fsw) * 8;
            const u16* as = As + stage * BM * BK + (wm * WM + frow) * BK + co;
            const u16* bs = Bs + stage * BN * BK + (wn * WN + frow) * BK + co;
#pragma unroll
            for (int j = 0; j < FM; ++j) xf[b][j] = *(const u16x8*)(as + j * F * BK);
#pragma unroll
            for (int i = 0; i < FN; ++i) wf[b][i] = *(const u16x8*)(bs + i * F * BK);
        };
        auto mfma_block = [&](auto bsel) {
            constexpr int b = decltype(bsel)::value;
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    if constexpr (F == 32) acc[i][j] = mfma_32x32x16(wf[b][i], xf[b][j], acc[i][j]);
                    else acc[i][j] = mfma_16x16x32(wf[b][i], xf[b][j], acc[i][j]);
                }
        };
        auto wait_landed = [&](int younger) {   // all but this wave's `younger` most recent tiles have landed + its ds_reads; barrier
            if (STAGES >= 4 && younger >= 3) wait_vm_lds_then_barrier<3 * PWT>();
            else if (STAGES >= 3 && younger == 2) wait_vm_lds_then_barrier<2 * PWT>();
            else if (younger == 1) wait_vm_lds_then_barrier<PWT>();
            else wait_vm_lds_then_barrier<0>();
        };
        typedef std::integral_constant<int, 0> B0;
        typedef std::integral_constant<int, 1> B1;
        static_assert(NKS % 2 == 0, "k-steps per K-tile");
        if (nkt > 0) {
            wait_landed(nkt - 1 < STAGES - 1 ? nkt - 1 : STAGES - 1);
            PCDM_STAMP(2);
            load_frags(0, 0, B0());
        }
        int cur = 0;
        for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
            for (int ks = 0; ks + 1 < NKS; ks += 2) {
                // the buffer-0 fragments were requested a whole MFMA block ago: this wait is free, and with at most one block of
                // reads outstanding the compiler's scoreboard stays exact (beyond 15 pending LDS operations it falls back to
                // lgkmcnt(0) in front of the MFMAs, which serialises the reads it was meant to overlap)
                wait_lds();
                load_frags(cur, ks + 1, B1());
                PCDM_SCHED_BARRIER();
                mfma_block(B0());
                PCDM_SCHED_BARRIER();
                if (ks + 2 < NKS) {
                    wait_lds();
                    load_frags(cur, ks + 2, B0());
                    PCDM_SCHED_BARRIER();
                    mfma_block(B1());
                    PCDM_SCHED_BARRIER();
                }
            }
            const int left = nkt - 2 - kt;   // tiles issued beyond kt + 1
            wait_landed(left < 0 ? 0 : (left < STAGES - 2 ? left : STAGES - 2));
            if (kt + STAGES < nkt && !(p.debug & 1)) issue_tile(kt + STAGES, cur);
            const int nx = cur + 1 == STAGES ? 0 : cur + 1;
            if (kt + 1 < nkt) load_frags(nx, 0, B0());
            PCDM_SCHED_BARRIER();
            mfma_block(B1());
            PCDM_SCHED_BARRIER();
            cur = nx;
        }
    }

    PCDM_STAMP(3);
    // ---- epilogue: lane holds, per (fn, fm, quad rg), channels n..n+3 of the pixel row (lane % F) of fragment row fm;
    // channel offset of quad rg inside its fragment: 8 rg + 4 (lane >> 5) for 32x32, 4 (lane >> 4) for 16x16
    const int half = lane >> LF;
    const int prow = lane & (F - 1);
    constexpr int QS = F == 32 ? 8 : 0;   // channel stride between the quads of one lane
    if (p.split_k > 1) {  // raw fp32 partial sums; bias / temb / residual are applied by the reduce kernel
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = m0 + wm * WM + j * F + prow;
            if (m >= p.M) continue;
            float* wr = p.ws + ((int64_t)ksplit * p.M + m) * p.Npad + n0 + wn * WN + 4 * half;
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int rg = 0; rg < NQ; ++rg) {
                    f32x4 v = {acc[i][j][4 * rg], acc[i][j][4 * rg + 1], acc[i][j][4 * rg + 2], acc[i][j][4 * rg + 3]};
                    *(f32x4*)(wr + i * F + QS * rg) = v;
                }
        }
        return;
    }
    // ---- LDS-staged epilogue (STORE / GEGLU): the accumulator quads (4 channels x 32 rows per instruction, i.e.
    // 16-byte pieces of 32 different 128-byte lines) go through a wave-private fp32 LDS tile and come back
    // row-major, so that every global access of the epilogue is a full 16 B per lane / whole 128-byte lines:
    // residual loads and bf16 stores.  Measured on the thin-K linears (K = 320): the direct quad stores
    // sustained only ~1.4 TB/s.  Single rounding is preserved (fp32 until the final convert).
    // GEGLU: the packed weight rows alternate [32 h | 32 gate], so a 64-wide wave tile holds 32 outputs: h in its fragment columns
    // [0, FN/2), the matching gates in [FN/2, FN)
    constexpr bool GLU_OK = WN == 64;
    const bool geglu = p.epilogue == PCDM_EPI_GEGLU;
    // q | k | v projections (PCDM_EPI_SPLIT_VT): tiles whose columns all lie below vt_col0 are plain stores
    // (per WAVE: the waves of one workgroup may take different paths when vt_col0 is not a multiple of BN -- the barrier below is
    //  therefore executed by every wave, before the paths part)
    const bool qk_tile = p.epilogue == PCDM_EPI_SPLIT_VT && n0 + wn * WN + WN <= p.vt_col0;
    const bool v_tile = p.epilogue == PCDM_EPI_SPLIT_VT && n0 + wn * WN >= p.vt_col0 && p.rows_per_batch % 32 == 0 && (p.ldo2 & 7) == 0 &&
                        p.act == 0 && !p.residual && !p.rowvec && p.M % 32 == 0;
    const bool lean = ((p.epilogue == PCDM_EPI_STORE && p.act == 0) || (geglu && GLU_OK) || qk_tile || v_tile) && (p.N & 7) == 0 &&
                      (p.ldo & 7) == 0 && (!p.residual || ((p.ldr & 7) == 0 && p.res_mod >= p.M)) && (!p.rowvec || p.rows_per_batch >= 32);
    __syncthreads();                                       // every wave is done with the operand stages
    if (lean) {
        // staged in passes of 32 pixel rows x <= 64 channels (one 128-byte line of bf16 per row): RB fragment rows x CG fragment columns
        constexpr int RB = 32 / F;                         // fragment rows per pass
        constexpr int CGM = 64 / F;                        // fragment columns per full pass
        constexpr int CG = FN < CGM ? FN : CGM;
        constexpr int EPW = CG * F + 4;                    // fp32 row pitch (conflict-free ds_write_b128)
        const bool has_res = p.residual != nullptr && !geglu, has_rv = p.rowvec != nullptr && !geglu;
        const int ncols_out = qk_tile ? p.vt_col0 : p.N;   // extent of a row of `out`
        // dup_rows (conv only): the whole epilogue runs a second time on the same accumulators for the output rows m + dup_rows -- with the
        // row-vector rows, residual rows and output rows of THAT half: the descriptors of repetition 1 are based dup_rows rows further,
        // every offset below stays relative to m (rows >= M are still dropped by the bounds check).  The CFG-shared prefix of the UNet:
        // both halves of the batch have the same input, so conv_in and the first resnet's conv1 are contracted once and written twice.
        auto epilogue_rep = [&](auto rep_tag) {
        constexpr int REP = decltype(rep_tag)::value;      // 0: rows m; 1 (conv, dup_rows > 0 only): rows m + dup_rows
        const int64_t sh = REP ? (int64_t)p.dup_rows : 0;
        const BufRsrc rs_o = make_buf_rsrc((const char*)p.out + sh * p.ldo * 2, (uint32_t)((((int64_t)p.M - 1) * p.ldo + ncols_out) * 2));
        const BufRsrc rs_r = make_buf_rsrc(has_res ? (const void*)(p.residual + sh * p.ldr) : (const void*)p.out,
                                           has_res ? (uint32_t)((((int64_t)p.M - 1) * p.ldr + p.N) * 2) : 0u);
        const float* rowvec0 = pcdm_gemm_detail::rowvec_base(p);
        const float* rowvec_rep = (has_rv && REP) ? rowvec0 + (sh / p.rows_per_batch) * p.ldrv : rowvec0;
        const BufRsrc rs_v = make_buf_rsrc(has_rv ? (const void*)rowvec_rep : (const void*)p.out,
                                           has_rv ? (uint32_t)((((int64_t)(p.M - 1) / p.rows_per_batch) * p.ldrv + p.N) * 4) : 0u);
        const BufRsrc rs_vt = make_buf_rsrc(v_tile ? (const void*)p.out2 : (const void*)p.out,
                                            v_tile ? (uint32_t)((int64_t)(p.M / p.rows_per_batch) * (p.N - p.vt_col0) * p.ldo2 * 2) : 0u);
        float* ep = (float*)smem + wave * (32 * EPW);      // wave-private 32 x (CG*F) tile
        // Everything the epilogue READS from global memory is fetched before its first store: a wave's vector-memory operations leave
        // its vmcnt counter in issue order, loads and stores alike, so a load issued behind a store (a bias quad per column group, a
        // row-vector quad per store instruction, a residual row two passes ahead -- the round-2 schedule) is handed over only after
        // that store has been acknowledged by the L2 / HBM: one exposed write round trip per pass (tools/gemm_anatomy.py: 22 k cycles
        // of epilogue with a residual, 10.5 k without).  Bias and the two time-embedding rows the wave tile can touch go to a
        // wave-private LDS slice, the residual rows of all passes to registers.
        constexpr int WNP = (WN + 3) / 4 * 4;
        float* bias_w = (float*)smem + NW * (32 * EPW) + wave * (3 * WNP);
        float* rvec_w = bias_w + WNP;
        const int wrow0 = m0 + wm * WM, wcol0 = n0 + wn * WN;
        const int rv_blo = wrow0 / p.rows_per_batch;
        const bool rv_lds = has_rv && (wrow0 + WM - 1) / p.rows_per_batch <= rv_blo + 1;
        const int rv_split_row = (rv_blo + 1) * p.rows_per_batch;
        {
            if (REP) PCDM_WAVE_SYNC();   // (the first repetition's reads of the wave's LDS slices are done)
            if (lane * 4 < WN) {
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                if (!REP) *(f32x4*)(bias_w + lane * 4) = p.bias ? *(const f32x4*)(p.bias + wcol0 + lane * 4) : z4;
                if (rv_lds) {
                    const int c = wcol0 + lane * 4;
                    const int nb = (p.M - 1) / p.rows_per_batch;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int b = rv_blo + h < nb ? rv_blo + h : nb;
                        f32x4 v = z4;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c + e < p.N) v[e] = rowvec_rep[(int64_t)b * p.ldrv + c + e];
                        *(f32x4*)(rvec_w + h * WNP + lane * 4) = v;
                    }
                }
            }
            PCDM_WAVE_SYNC();
        }
        constexpr int NCG = (FN + CG - 1) / CG, NJB = WM / 32;
        // pass q = (column group q / NJB, row block q % NJB); GEGLU needs WN == 64, i.e. a single column group
        // The pass loop exists twice: with and without a residual operand.  The residual rows of ALL passes sit in registers from before
        // the first store (96 VGPRs for a 96x80 wave tile); the instance without a residual does not carry them (a single instance
        // with a run-time flag spilled into scratch, whose reloads are vector-memory loads behind stores again).
        auto run_passes = [&](auto res_tag) {
        constexpr bool RES = decltype(res_tag)::value;
        u32x4 rv[RES ? NCG * NJB : 1][4];                  // residual rows of every pass (static indices: registers)
        auto pass_cols = [&](int q) { const int i0 = (q / NJB) * CG; return geglu ? 32 : ((FN - i0) < CG ? (FN - i0) : CG) * F; };
        auto pass_ncol0 = [&](int q) { return geglu ? (n0 + wn * WN) / 2 : n0 + wn * WN + (q / NJB) * CG * F; };
        auto issue_res = [&](int q) {
            if constexpr (!RES) {
#pragma unroll
                for (int it = 0; it < 4; ++it) rv[0][it] = u32x4{0, 0, 0, 0};
                return;
            }
            const int mrow0 = m0 + wm * WM + (q % NJB) * 32, wc = pass_cols(q), nc = pass_ncol0(q);
            if (wc == 64) lean_res_load<64>(p, lane, mrow0, nc, rs_r, rv[q]);
            else if (wc == 32) lean_res_load<32>(p, lane, mrow0, nc, rs_r, rv[q]);
            else if (wc == 16) lean_res_load<16>(p, lane, mrow0, nc, rs_r, rv[q]);
            else lean_res_load<48>(p, lane, mrow0, nc, rs_r, rv[q]);
        };
        // residual rows of ALL passes -> registers, before the first store (see above); already there for the small wave tiles
        if constexpr (EARLY_RES && RES) {
            static_assert(E_NPASS == NCG * NJB, "pass geometry");
            if (early_res) {
#pragma unroll
                for (int q = 0; q < NCG * NJB; ++q)
#pragma unroll
                    for (int it = 0; it < 4; ++it) rv[q][it] = rv_early[q][it];
            } else {
#pragma unroll
                for (int q = 0; q < NCG * NJB; ++q) issue_res(q);
            }
        } else {
#pragma unroll
            for (int q = 0; q < (RES ? NCG * NJB : 1); ++q) issue_res(q);
        }
        const bool swiglu = p.act == PCDM_ACT_SILU;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;          // bias of this lane's 8 channels: re-read (LDS) once per column group
#pragma unroll
        for (int q = 0; q < NCG * NJB; ++q) {
            const int i0 = (q / NJB) * CG, jb = q % NJB;
            const int ng = (FN - i0) < CG ? (FN - i0) : CG;   // fragment columns in this pass (the last column group may be narrower)
            const int ncol0 = pass_ncol0(q), wc = pass_cols(q);
            if (jb == 0 && !geglu && !v_tile) {               // (GEGLU applies its biases before the gate, the V^T pass per channel)
                const int c8_ = (lane % (wc / 8)) * 8;
                const float* bp = bias_w + (ncol0 - wcol0) + c8_;      // (channels >= N hold the packed weights' zero padding)
                b0 = *(const f32x4*)bp;
                b1 = *(const f32x4*)(bp + 4);
            }
            // 1. quads -> LDS [row = pixel][col = channel]
            if (geglu) {
                if constexpr (GLU_OK) {
#pragma unroll
                    for (int jj = 0; jj < RB; ++jj)
#pragma unroll
                        for (int i = 0; i < FN / 2; ++i)
#pragma unroll
                            for (int rg = 0; rg < NQ; ++rg) {
                                const int nl = i * F + QS * rg + 4 * half;   // 0..31 within the wave's 32 outputs
                                const f32x4 bh = *(const f32x4*)(bias_w + nl);
                                const f32x4 bg = *(const f32x4*)(bias_w + nl + 32);
                                const acc_t& ah = acc[i][jb * RB + jj];
                                const acc_t& ag = acc[i + FN / 2][jb * RB + jj];
                                f32x4 v;
                                if (swiglu) {   // (one wave-uniform branch per quad: with gate_act(., p.act) hipcc branched per element)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = (ah[4 * rg + e] + bh[e]) * silu_f(ag[4 * rg + e] + bg[e]);
                                } else {
                                    const f32x4 hq = {ah[4 * rg], ah[4 * rg + 1], ah[4 * rg + 2], ah[4 * rg + 3]};
                                    const f32x4 gq = {ag[4 * rg], ag[4 * rg + 1], ag[4 * rg + 2], ag[4 * rg + 3]};
                                    v = geglu_quad(hq + bh, gq + bg);
                                }
                                *(f32x4*)(ep + (jj * F + prow) * EPW + nl) = v;
                            }
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < RB; ++jj)
#pragma unroll
                    for (int i = 0; i < CG; ++i) {
                        if (i0 + i >= FN) break;
#pragma unroll
                        for (int rg = 0; rg < NQ; ++rg) {
                            const acc_t& a_ = acc[i0 + i < FN ? i0 + i : 0][jb * RB + jj];
                            const f32x4 v = {a_[4 * rg], a_[4 * rg + 1], a_[4 * rg + 2], a_[4 * rg + 3]};
                            *(f32x4*)(ep + (jj * F + prow) * EPW + i * F + QS * rg + 4 * half) = v;
                        }
                    }
            }
            PCDM_WAVE_SYNC();
            // 2. row-major read-back (same wave: LDS ops complete in order), 16-byte buffer stores
            const int mrow0 = m0 + wm * WM + jb * 32;
            if (v_tile) {
                if (mrow0 < p.M) {
                    const float* bc = bias_w + (ncol0 - wcol0);
                    if (ng * F == 64) vt_pass<64, EPW>(p, ep, lane, mrow0, ncol0, rs_vt, bc);
                    else if (ng * F == 32) vt_pass<32, EPW>(p, ep, lane, mrow0, ncol0, rs_vt, bc);
                    else if (ng * F == 16) vt_pass<16, EPW>(p, ep, lane, mrow0, ncol0, rs_vt, bc);
                    else vt_pass<48, EPW>(p, ep, lane, mrow0, ncol0, rs_vt, bc);
                }
            } else {
                const int rc0 = ncol0 - wcol0;
#define PCDM_LEAN(W, RVK) lean_pass<W, EPW, RVK>(p, ep, lane, mrow0, ncol0, b0, b1, rs_o, rs_v, rv[RES ? q : 0], rvec_w, WNP, rc0, rv_split_row)
#define PCDM_LEAN_W(RVK)                 \
    do {                                 \
        if (wc == 64) PCDM_LEAN(64, RVK);      \
        else if (wc == 32) PCDM_LEAN(32, RVK); \
        else if (wc == 16) PCDM_LEAN(16, RVK); \
        else PCDM_LEAN(48, RVK);               \
    } while (0)
                if (!has_rv) PCDM_LEAN_W(0);
                else if (rv_lds) PCDM_LEAN_W(1);
                else PCDM_LEAN_W(2);
#undef PCDM_LEAN_W
#undef PCDM_LEAN
            }
            PCDM_WAVE_SYNC();   // this pass's reads precede the next pass's writes
        }
        };
        if (has_res) run_passes(std::true_type());
        else run_passes(std::false_type());
        };   // epilogue_rep
        epilogue_rep(std::integral_constant<int, 0>());
        if constexpr (CONV) {   // (the second half of a dup_rows launch: its own output / residual / row-vector rows, the same accumulators)
            if (p.dup_rows > 0) epilogue_rep(std::integral_constant<int, 1>());
        }
#ifndef PCDM_EMU
        if ((p.debug & 4) && p.ws) {
            PCDM_STAMP(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the CU
            PCDM_STAMP(5);
            if (lane == 0) {
                unsigned long long* o = (unsigned long long*)p.ws + ((int64_t)blockIdx.x * NW + wave) * 8;
                for (int i = 0; i < 6; ++i) o[i] = stamp[i];
                o[6] = wg;
            }
        }
#endif
        return;
    }
    const float* rowvec_g = pcdm_gemm_detail::rowvec_base(p);
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = m0 + wm * WM + j * F + prow;
        if (m >= p.M) continue;
        const int bidx = m / p.rows_per_batch;
        const int tok = m - bidx * p.rows_per_batch;
        const int64_t rrow = p.residual ? (int64_t)(m % p.res_mod) * p.ldr : 0;
        if (p.epilogue == PCDM_EPI_GEGLU) {
            if constexpr (GLU_OK) {
#pragma unroll
                for (int i = 0; i < FN / 2; ++i)
#pragma unroll
                    for (int rg = 0; rg < NQ; ++rg) {
                        const int nl = i * F + QS * rg + 4 * half;  // 0..31 within the wave's 32 outputs
                        const int nh = n0 + wn * WN + nl;           // packed row of h
                        const int ng = nh + 32;                     // packed row of gate
                        const int no = (n0 + wn * WN) / 2 + nl;     // output channel
                        if (no >= p.N) continue;
                        const f32x4 bh = *(const f32x4*)(p.bias + nh), bg = *(const f32x4*)(p.bias + ng);
                        u16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float hval = acc[i][j][4 * rg + e] + bh[e];
                            const float gval = acc[i + FN / 2][j][4 * rg + e] + bg[e];
                            o[e] = f2bf(hval * gate_act(gval, p.act));
                        }
                        *(u16x4*)((u16*)p.out + (int64_t)m * p.ldo + no) = o;
                    }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < FN; ++i) {
#pragma unroll
            for (int rg = 0; rg < NQ; ++rg) {
                const int n = n0 + wn * WN + i * F + QS * rg + 4 * half;
                if (n >= p.N) continue;
                // unconditional loads (absent operands read zeros): see g_zero32
                const f32x4 bv = *(const f32x4*)(p.bias ? p.bias + n : (const float*)g_zero32);
                const f32x4 tv = *(const f32x4*)(rowvec_g ? rowvec_g + (int64_t)bidx * p.ldrv + n : (const float*)g_zero32);
                const u16x4 rv = *(const u16x4*)(p.residual ? p.residual + rrow + n : (const u16*)g_zero32);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * rg + e] + bv[e] + tv[e];
                if (p.act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
                if (p.epilogue == PCDM_EPI_NCHW_F32) {
                    float* o = (float*)p.out;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) o[((int64_t)bidx * p.N + n + e) * p.rows_per_batch + tok] = v[e];
                } else if (p.epilogue == PCDM_EPI_SPLIT_VT && n >= p.vt_col0) {
                    const int cv = p.N - p.vt_col0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        p.out2[((int64_t)bidx * cv + (n + e - p.vt_col0)) * p.ldo2 + tok] = f2bf(v[e]);
                } else {
                    u16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
                    *(u16x4*)((u16*)p.out + (int64_t)m * p.ldo + n) = o;
                }
            }
        }
    }
}

// out[m, n..n+3] = epilogue( sum_s ws[s][m][n..n+3] ); one thread per (row, channel quad); PCDM_EPI_STORE only
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const int nq = p.N / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.M * nq) return;
    const int m = (int)(i / nq), n = (int)(i - (int64_t)m * nq) * 4;
    // epilogue operands first (unconditional loads; absent ones read zeros, see g_zero32): they return while the slabs are summed
    const f32x4 bv = *(const f32x4*)(p.bias ? p.bias + n : (const float*)g_zero32);
    const float* rowvec_g = pcdm_gemm_detail::rowvec_base(p);
    const f32x4 tv = *(const f32x4*)(rowvec_g ? rowvec_g + (int64_t)(m / p.rows_per_batch) * p.ldrv + n : (const float*)g_zero32);
    const u16x4 rv = *(const u16x4*)(p.residual ? p.residual + (int64_t)(m % p.res_mod) * p.ldr + n : (const u16*)g_zero32);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    // slabs summed in a fixed order, four independent 16-byte loads in flight at a time (a one-load-per-iteration loop
    // pays one L2 / HBM round trip per slab)
    const float* wp = p.ws + (int64_t)m * p.Npad + n;
    const int64_t slab = (int64_t)p.M * p.Npad;
    int s = 0;
    for (; s + 4 <= p.split_k; s += 4) {
        const f32x4 t0 = *(const f32x4*)(wp + (s + 0) * slab), t1 = *(const f32x4*)(wp + (s + 1) * slab);
        const f32x4 t2 = *(const f32x4*)(wp + (s + 2) * slab), t3 = *(const f32x4*)(wp + (s + 3) * slab);
        v += t0;
        v += t1;
        v += t2;
        v += t3;
    }
    for (; s < p.split_k; ++s) v += *(const f32x4*)(wp + s * slab);
    v += bv;
    v += tv;
    if (p.act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
    u16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
    *(u16x4*)((u16*)p.out + (int64_t)m * p.ldo + n) = o;
}

template <int BM, int BN, int WGM, int WGN, int STAGES, bool CONV, bool STAG = false, int F = 32, int KB = 64>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
    constexpr int BK = KB;
    // operand ring, or the wave-private fp32 epilogue tiles (32 x (WN + 4) floats per wave) if those need more
    constexpr int smem_ops = STAGES * (BM + BN) * BK * (int)sizeof(u16);
    constexpr int FN_ = BN / WGN / F, CGM_ = 64 / F;
    constexpr int WNP_ = (BN / WGN + 3) / 4 * 4;   // + per wave: bias slice and two row-vector rows (3 x WN floats)
    constexpr int smem_epi = WGM * WGN * (32 * ((FN_ < CGM_ ? FN_ : CGM_) * F + 4) + 3 * WNP_) * (int)sizeof(float);
    constexpr int smem = smem_ops > smem_epi ? smem_ops : smem_epi;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, WGM, WGN, STAGES, CONV, STAG, F, KB>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    if (a.Npad % BN) return -1;                                        // the N tiles must cover Npad exactly
    if (a.epilogue == PCDM_EPI_GEGLU && BN / WGN != 64) return -1;   // GEGLU pairs [32 h | 32 gate] need a 64-wide wave tile
    GemmArgs g = a;
    g.tiles_m = (a.M + BM - 1) / BM;
    g.tiles_n = a.Npad / BN;
    PCDM_LAUNCH(PCDM_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, STAGES, CONV, STAG, F, KB>), dim3(g.tiles_m * g.tiles_n * g.split_k),
                dim3(WGM * WGN * 64), smem, st, g);
    PCDM_CHECK_LAUNCH();
    if (g.split_k > 1 && !g.defer_reduce) {
        const int64_t n = (int64_t)g.M * (g.N / 4);
        PCDM_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g);
        PCDM_CHECK_LAUNCH();
    }
    return 0;
}

// Tile configurations (id: BM x BN, waves, LDS stages -> LDS bytes, resident blocks per CU).
// Which one wins depends on M, N, K and on how many workgroups the problem yields (measured: tools/bench_ops.py);
// pcdms_amd.ops autotunes per problem shape at warm-up, id 0 = the static heuristic below.
template <bool CONV>
int dispatch_tile(int tile, const GemmArgs& a, hipStream_t st) {
    switch (tile) {
#ifndef PCDM_DEV_NEW_TILES_ONLY   // (developer builds of a few instantiations: tools/ubench; never defined for the product)
        case 1: return launch_gemm<256, 128, 4, 2, 3, CONV>(a, st);   // 8 waves, 144 KiB, 1 block / CU
        case 2: return launch_gemm<64, 64, 2, 2, 2, CONV>(a, st);     // 4 waves,  32 KiB, 4+ blocks / CU
        case 3: return launch_gemm<256, 64, 4, 2, 2, CONV>(a, st);    // 8 waves,  80 KiB, 2 blocks / CU
        case 4: return launch_gemm<128, 128, 2, 2, 2, CONV>(a, st);   // 4 waves,  64 KiB, 2 blocks / CU
        case 5: return launch_gemm<128, 64, 2, 2, 2, CONV>(a, st);    // 4 waves,  48 KiB, 3 blocks / CU
        case 6: return launch_gemm<256, 64, 4, 2, 3, CONV>(a, st);    // 8 waves, 120 KiB, 1 block / CU
        case 7: return launch_gemm<128, 128, 2, 2, 3, CONV>(a, st);   // 4 waves,  96 KiB, 1 block / CU
        case 8: return launch_gemm<64, 64, 2, 2, 4, CONV>(a, st);     // 4 waves,  64 KiB, 2 blocks / CU
        case 9: return launch_gemm<256, 128, 4, 2, 2, CONV>(a, st);   // 8 waves,  96 KiB, 1 block / CU
        case 10: return launch_gemm<128, 64, 2, 2, 3, CONV>(a, st);   // 4 waves,  72 KiB, 2 blocks / CU
        case 11: return launch_gemm<256, 128, 4, 2, 3, CONV, true>(a, st);  // as 1, staggered wave groups
        case 12: return launch_gemm<256, 64, 4, 2, 3, CONV, true>(a, st);   // as 6, staggered wave groups
        // N-narrow tiles whose waves still own 64x64 (4 fragment loads per 4 MFMAs instead of 3 per 2: the 64x32 wave
        // tiles of 3/5/6 run at ~75 % of the LDS read bandwidth) and whose epilogue writes whole 128-byte lines
        case 13: return launch_gemm<256, 64, 4, 1, 2, CONV>(a, st);   // 4 waves,  80 KiB, 2 blocks / CU
        case 14: return launch_gemm<256, 64, 4, 1, 3, CONV>(a, st);   // 4 waves, 120 KiB, 1 block / CU
        case 15: return launch_gemm<128, 64, 2, 1, 2, CONV>(a, st);   // 2 waves,  48 KiB, 3 blocks / CU
        case 16: return launch_gemm<512, 64, 8, 1, 2, CONV>(a, st);   // 8 waves, 144 KiB, 1 block / CU
        // 256x256: 8 waves x (128x64): 32 MFMAs per wave per K-tile for the same 8 LDS-DMA instructions (1:4 instead of 1:2.7)
        case 17: return launch_gemm<256, 256, 2, 4, 2, CONV>(a, st);  // 8 waves, 128 KiB, 1 block / CU; N % 256 == 0
        // 128x128 with EIGHT waves (32x64 each): the same 64 KiB and 2 blocks / CU as tile 4, but 4 waves per SIMD to hide
        // LDS / MFMA-result latency (the 4-wave tiles run 2 waves per SIMD)
        case 18: return launch_gemm<128, 128, 4, 2, 2, CONV>(a, st);
#endif
        // Full-row tiles for N = 320 k (every UNet level: 320 / 640 / 960 / 1280 / 1920 / 3840 channels): BN = 320, wave tile
        // 64..128 x 160.  With BN = 64 the A tile is re-fetched by 5 N tiles and the loop sits on the L2 -> LDS rate
        // (~19 TB/s measured) and the LDS read rate (1.5 ds_read_b128 per MFMA); a 64x160 wave tile needs 0.7 reads per MFMA
        // and 2.8x fewer staged bytes per FLOP.
        // (first built with 32x32 fragments -- 256x320 / 8 waves of 64x160, 128x320, 128x160 -- which ran the loop at ~1.28 PF/s
        //  but left 80 of the 256 CUs idle at M = 45056 (176 tiles); superseded by the 16x16x32-fragment tiles below)
        // (192x320 / 256x320 with FOUR waves of 96x160 / 128x160 32x32 fragments -- one wave per SIMD, AGPR accumulators -- were tried
        //  and dropped: hipcc 7.2 crashes in 'AMDGPU Rewrite AGPR-Copy-MFMA' under -amdgpu-mfma-vgpr-form=1 and spills ~2 KiB per
        //  lane without it)
        // 16x16x32 fragments: wave tile 96x80.  192-row tiles: M = 45056 -> 235 workgroups (N = 320), M = 11264 -> 59 x 2 (N = 640),
        // M = 2816 -> 15 x 4 (N = 1280); 96-row tiles: M = 11264 -> 118 x 2 = 236 -- all within 8 % of the 256 CUs.
        case 21: return launch_gemm<192, 320, 2, 4, 2, CONV, false, 16>(a, st);   // 8 waves, 128 KiB, 1 block / CU
        // (measured and dropped, never selected by the tuner: 128x320 / 4 waves; 96x320 / 4 waves of 96x80 with two or three stages;
        //  128x160 / 2 waves with three stages)
        case 26: return launch_gemm<192, 256, 2, 4, 2, CONV, false, 16>(a, st);   // 8 waves (96x64 each: GEGLU-capable), 112 KiB
        // (measured and dropped, never selected by the tuner: tile 21 / a 256x256 tile with FOUR stages of K = 32 -- three tiles in flight
        //  in the same 128 KiB, template parameter KB = 32 -- 10-15 % slower: 64-byte rows double the cache-line requests per staged
        //  byte, and the loop was not latency-bound in the first place: tools/ablate_gemm.py, profiles/r2_gemm_ablation.txt)
#ifdef PCDM_DEV_KB32
        case 27: return launch_gemm<192, 320, 2, 4, 4, CONV, false, 16, 32>(a, st);
        case 28: return launch_gemm<256, 256, 2, 4, 4, CONV, false, 16, 32>(a, st);
#endif
        default: return -1;
    }
}
}  // namespace

extern "C" int pcdm_gemm(const pcdm_gemm_params* p, pcdm_stream_t s) {
    if (!p || !p->a || !p->w || !p->out) return -1;
    if (p->M <= 0 || p->N <= 0 || p->K <= 0 || p->K % BK || p->Npad % 64 || p->Npad < p->N || p->N % 4) return -1;
    if (p->rows_per_batch <= 0) return -1;
    // 16-byte vector loads of the epilogue operands: bias and rowvec bases, the row pitch and the per-step block stride
    if (((uintptr_t)p->bias & 15) || (p->rowvec && (((uintptr_t)p->rowvec & 15) || (p->ldrv > 0 && p->ldrv % 4) || (p->rowvec_step && p->rowvec_step_stride % 4))))
        return -1;
    // 32-bit buffer offsets: every operand must stay below 2 GiB
    const int64_t lim = 0x7fffffffLL;
    if ((int64_t)p->Npad * (p->ldw > 0 ? p->ldw : p->K) * 2 >= lim) return -2;
    if (p->conv ? ((int64_t)p->B * p->Hi * p->Wi * p->cin * 2 + ((int64_t)p->Wi + 1) * p->cin * 2 >= lim)
                : ((int64_t)p->M * p->lda * 2 >= lim || (p->a2 && (int64_t)p->M * p->lda2 * 2 >= lim))) return -2;
    const int64_t out_esz = p->epilogue == PCDM_EPI_NCHW_F32 ? 4 : 2;   // (fp32 only there; the x4 applied to every epilogue refused the
                                                                         //  VAE decoder's 2.9 M x 256 bf16 upsampling convs at 8 samples)
    if ((int64_t)p->M * (p->ldo > 0 ? p->ldo : p->N) * out_esz >= lim || (p->residual && (int64_t)p->M * (p->ldr > 0 ? p->ldr : p->N) * 2 >= lim))
        return -2;   // (32-bit offsets in the epilogue's buffer stores / loads)
    GemmArgs a;
    a.a = (const u16*)p->a;
    a.a2 = (const u16*)p->a2;
    a.lda = p->lda;
    a.lda2 = p->lda2;
    a.c1 = p->a2 ? p->c1 : p->K;
    a.B = p->B; a.Hi = p->Hi; a.Wi = p->Wi; a.Ho = p->Ho; a.Wo = p->Wo;
    a.stride = p->stride; a.upsample = p->upsample; a.cin = p->cin;
    a.pad = p->no_pad_lo ? 0 : 1;
    a.w = (const u16*)p->w;
    a.ldw = p->ldw > 0 ? p->ldw : p->K;
    if (a.ldw < p->K || a.ldw % 8) return -1;
    a.M = p->M; a.N = p->N; a.K = p->K; a.Npad = p->Npad;
    a.bias = p->bias;
    a.rowvec = p->rowvec;
    a.rowvec_step = p->rowvec ? p->rowvec_step : nullptr;
    a.rowvec_step_stride = p->rowvec_step_stride;
    a.ldrv = p->ldrv > 0 ? (int)p->ldrv : p->N;
    a.rows_per_batch = p->rows_per_batch;
    a.residual = (const u16*)p->residual;
    a.ldr = p->ldr;
    a.res_mod = p->res_mod > 0 ? p->res_mod : p->M;
    a.epilogue = p->epilogue;
    a.vt_col0 = p->vt_col0;
    a.out = p->out;
    a.ldo = p->ldo;
    a.out2 = (u16*)p->out2;
    a.ldo2 = p->ldo2;
    a.tiles_m = a.tiles_n = 0;
    a.debug = p->tile >> 8;
    a.act = p->act;
    a.dup_rows = p->dup_rows;
    if (a.dup_rows) {   // the lean (LDS-staged) epilogue of a convolution only: everything it needs is known here
        if (a.dup_rows < 0 || !p->conv || p->epilogue != PCDM_EPI_STORE || p->act || p->split_k > 1 || (p->N & 7) || (p->ldo & 7)) return -1;
        if (p->residual && ((p->ldr & 7) || (p->res_mod > 0 && p->res_mod < p->M))) return -1;
        if (p->rowvec && (p->rows_per_batch < 32 || a.dup_rows % p->rows_per_batch)) return -1;
        if (((int64_t)p->M + a.dup_rows) * (p->ldo > 0 ? p->ldo : p->N) * 2 >= lim) return -2;
    }
    a.zero_rows = p->zero_rows;
    if (a.zero_rows < 0 || a.zero_rows > p->M || (a.zero_rows && p->conv)) return -1;
    if (a.act < 0 || a.act > PCDM_ACT_GELU || (a.act == PCDM_ACT_GELU && p->epilogue == PCDM_EPI_GEGLU)) return -1;
    a.ln_wsum = p->ln_wsum;
    a.ln_eps = p->ln_eps;
    a.split_k = p->split_k > 1 ? p->split_k : 1;
    a.defer_reduce = (p->defer_reduce && a.split_k > 1) ? 1 : 0;
    if (a.defer_reduce && (p->act || p->res_mod > 0 && p->res_mod < p->M)) return -1;   // (the consumer applies bias / rowvec / residual only)
    a.ws = p->ws;
    if (a.split_k > 1) {
        if (p->epilogue != PCDM_EPI_STORE || !p->ws || a.split_k > p->K / BK || a.split_k > 64) return -1;
        if (p->ws_floats < (int64_t)a.split_k * p->M * p->Npad) return -1;
    }
    if (p->conv) {
        if (p->cin % BK || p->K != 9 * p->cin || (p->stride != 1 && p->stride != 2) || p->a2) return -1;
        if (p->upsample && (p->stride != 1 || p->no_pad_lo || p->Ho < p->Hi || p->Wo < p->Wi)) return -1;
        if (p->M != p->B * p->Ho * p->Wo) return -1;
    } else {
        if (p->a2 && (p->c1 % BK || p->c1 <= 0 || p->c1 >= p->K)) return -1;
    }
    if (p->epilogue == PCDM_EPI_GEGLU && (!p->bias || p->Npad % 128 || p->N * 2 > p->Npad)) return -1;
    if (p->epilogue == PCDM_EPI_SPLIT_VT && (!p->out2 || p->vt_col0 % 4)) return -1;
    hipStream_t st = (hipStream_t)s;
    int tile = p->tile & 0xff;
    if (tile >= pcdm_gemm_detail::kRowGemmTile0) {
        if ((a.debug & 4) && p->ws_floats < (int64_t)((p->M + 95) / 96) * 8 * 8 * 2) return -1;   // (stamps: 8 x uint64 per wave)
        return p->conv ? -1 : pcdm_gemm_detail::launch_rowgemm(tile, a, st);
    }
    if (a.ln_wsum) return -1;   // the folded LayerNorm needs the row statistics: the A-in-registers kernel only
    const bool n128 = p->Npad % 128 == 0;
    const bool needs128 = tile == 1 || tile == 4 || tile == 7 || tile == 9 || tile == 11 || tile == 18;
    if (p->epilogue == PCDM_EPI_GEGLU && tile != 0 && !needs128 && tile < 13) return -1;  // GEGLU pairs need a 64-wide wave tile (19+: launch_gemm checks)
    if (tile == 0) {
        const int64_t t256 = (int64_t)((p->M + 255) / 256) * (p->Npad / 128);
        const int64_t t128 = (int64_t)((p->M + 127) / 128) * (p->Npad / 128);
        if (n128 && t256 >= 200) tile = 1;
        else if (n128 && (t128 >= 160 || p->epilogue == PCDM_EPI_GEGLU)) tile = 4;
        else if (!n128 && (int64_t)((p->M + 127) / 128) * (p->Npad / 64) >= 256) tile = 5;
        else tile = 2;
    } else if ((needs128 && !n128) || (tile == 17 && p->Npad % 256)) {
        return -1;
    }
    return p->conv ? dispatch_tile<true>(tile, a, st) : dispatch_tile<false>(tile, a, st);
}
