// Fused (flash-style) attention for head_dim 64 on gfx950 (SURVEY.md §2.1 K9 self-attention,
// K10 cross-attention).  Replaces xformers.ops.memory_efficient_attention
// (stage2_batchtest_inpaint_model.py:133): o = softmax(q k^T * scale) v, fp32 softmax, no mask.
//
// Design (wave = 64, v_mfma_f32_32x32x16_bf16):
//  * workgroup = 4 waves; each wave owns 32 query rows and the whole head_dim; K / V^T tiles of 64 keys go
//    HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds: constant per-lane offsets + a wave-uniform SGPR offset per
//    tile, out-of-range keys zero-filled by the hardware bounds check), double-buffered, one barrier per tile.
//  * QK^T is issued "swapped": S^T = K * Q^T, so a lane holds 32 scores of ONE query (the other 32
//    live in lane^32).  Row max / row sum are 31 in-register ops + one __shfl_xor(.,32); the online
//    softmax rescale of O^T is lane-uniform.  No serial-lane softmax, no LDS round trip for P.
//  * The K rows fed to MFMA row i are keys pi(i) (bits 2 and 3 of i swapped).  With that
//    permutation the 8 accumulator registers r = 8s..8s+7 of a lane are exactly the 8 consecutive
//    keys the lane must supply as the B operand of O^T += V^T * P^T, so P goes from accumulator to
//    MFMA operand with a bf16 convert only (no permlane / ds_bpermute).
//  * V is consumed as V^T [d][key] (key contiguous): the projection GEMM writes it transposed
//    (PCDM_EPI_SPLIT_VT), so the A operand of the PV MFMA is a plain ds_read_b128 -- no transpose
//    anywhere in this kernel.
//  * LDS rows are unpadded and XOR-swizzled (conflict-free ds_read_b128, see gemm.hip).
//  * The softmax is VALU-bound at head_dim 64 (one exp per 256 FLOP), so the VALU work per score is cut to the exp itself:
//    Q is pre-multiplied by scale * log2(e) (once per workgroup), and the running reference m of each query is subtracted INSIDE the
//    QK^T contraction by one extra MFMA per key fragment (A = ones, B = a fragment holding bf16(-m) in its k = 0 slot), so that
//    P = exp2(S') with S' straight out of the accumulator -- no per-score fma.  m is LAZY (cdna_hip_programming.md T13): it moves
//    only when a tile's maximum exceeds it by more than `thr` (log2 units; the first tile always sets it), and only then are the
//    scores of that tile, O and the row sum rescaled.  m is kept bf16-exact, so the value the MFMA subtracts is exactly the one the
//    rescale factors are computed from; any reference would do for the mathematics (P, the row sum and O all carry the same
//    2^-m), thr only bounds P <= 2^thr.  thr = 0 reproduces the eager online softmax.
#include "pcdm_device.h"
#include "../../include/pcdm.h"

#include <stdlib.h>

#include <type_traits>

namespace {
constexpr int KB = 64;     // keys per tile
constexpr int QPW = 32;    // queries per wave
constexpr int QPB = 128;   // queries per workgroup

// XCD-aware placement of the (query block, head, batch) grid (round 6).  The hardware deals workgroups to the 8 XCDs round robin by their
// linear id, so the 44 query blocks of one (batch, head) at N = 5632 -- which all stream the SAME 1.4 MB of K / V^T -- used to land on all
// eight L2s: every XCD fetched every head's K / V^T across the fabric (rocprofv3 FETCH_SIZE, profiles/r5_kernel_traffic.json: 4.6 GB per
// step for the 32 attention launches, ~ 850 MB per level-0 self-attention launch against 86 MB of q + k + v).  The bijection below (the
// same one gemm_kernel uses) hands XCD x the x-th CONTIGUOUS eighth of the virtual ids, query block fastest: the ~ 96 workgroups resident
// on an XCD then cover 2-3 heads, whose K / V^T (3-4 MB) live in that XCD's 4 MB L2.  flags bit 0 = 0: the plain grid (A/B switch).
__device__ __forceinline__ void attn_block_coords(int flags, int& qb, int& h, int& b) {
    if (!(flags & 1)) { qb = blockIdx.x; h = blockIdx.y; b = blockIdx.z; return; }
    const int nqb = gridDim.x, H = gridDim.y;
    const int total = nqb * H * (int)gridDim.z;
    const int lin = blockIdx.x + nqb * (blockIdx.y + H * blockIdx.z);
    const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    const int bh = v / nqb;
    qb = v - bh * nqb;
    b = bh / H;
    h = bh - b * H;
}

// (Round 4 measured a software-pipelined form of this kernel -- the QK^T MFMAs of tile t + 1 issued in one scheduling region with the
//  exponentials of tile t, [1 MFMA : 4 v_exp + 2 v_cvt_pk] placed by a sched_group_barrier pipeline, K one tile ahead of V^T in the ring, 32
//  more accumulator registers => two workgroups per CU: 786-824 TF/s at N = 5632 against this kernel's 852-892 in the same session
//  (profiles/r4_bench_attn_pipe_ab.txt).  Three independent waves per SIMD cover each other's softmax better than a wave covers its own.
//  A FOUR-workgroups-per-CU instance of this kernel (fragments read in halves, V^T requested behind the exponentials, 128 VGPRs) spilled
//  80-112 B per lane and ran at 630 TF/s (profiles/r4_bench_attn_lowreg_ab.txt): the 166 registers of this form are its working set.)
// ROWSUM_VALU: the softmax denominator as per-lane fp32 adds of the un-rounded P (combined across the two lane halves once, at the
// end) instead of an MFMA against a ones fragment (4 of the 22 MFMAs per tile); which one wins depends on which pipe has slack.
// (s_setprio 1 around the two MFMA clusters was measured too: no gain with 3 co-resident waves per SIMD -- removed.)
// (Round 5 also measured the cross-attention with its query path inside the kernel -- norm2 -> attn2.to_q on the matrix pipe straight from
//  global memory in front of the key loop -- 3-40 % SLOWER than the launches it replaced (profiles/r5_bench_xattn.txt); removed in round 6,
//  the code is in the git history: commit 'cross-attention as one launch'.)
template <bool ROWSUM_VALU>
__global__ __launch_bounds__(256, 3) void flash_attn_kernel(const u16* __restrict__ q, int64_t ldq,
                                                         const u16* __restrict__ k, int64_t ldk,
                                                         const u16* __restrict__ vt, int64_t ldvt,
                                                         u16* __restrict__ o, int64_t ldo, int H, int Lq, int Lk,
                                                         float c /* scale * log2(e) */, float thr /* lazy-rescale threshold, log2 units */, int flags) {
    // K tile [64 keys][64 d] and V^T tile [64 d][64 keys], 2 stages each, unpadded 128-byte rows whose 16-byte
    // chunks are XOR-swizzled by (row>>1)&7 (applied on the DMA source offset and on the fragment reads, exactly
    // as in gemm.hip): conflict-free ds_read_b128, filled by buffer_load ... lds with no VGPR round trip.
    // (ONE __shared__ object: with a second one hipcc drains vmcnt before the first ds_read after a DMA issue)
    __shared__ __attribute__((aligned(16))) u16 KV[2][2][KB * 64];   // [stage][K | V^T][row * 64 + col]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int qb, h, b;
    attn_block_coords(flags, qb, h, b);
    const int q0 = qb * QPB + wave * QPW;
    const int hh = lane >> 5, col = lane & 31;

    // Q fragments (B operand of S^T = K Q^T): lane -> query col, d = 16*ks + 8*hh + e
    int qrow = q0 + col;
    const bool qvalid = qrow < Lq;
    if (!qvalid) qrow = Lq - 1;
    u16x8 qf[4];
    const int pi = (col & 0x13) | ((col & 4) << 1) | ((col & 8) >> 1);  // K row permutation
    {
        const u16* qp = q + ((int64_t)b * Lq + qrow) * ldq + h * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const u16x8 raw = *(const u16x8*)(qp + ks * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = f2bf(bf2f(raw[e]) * c);   // scores come out of the MFMA in log2 units
        }
    }

    // LDS-DMA staging: wave w, instruction j fills rows (2w+j)*8 .. +7 of the K tile and of the V^T tile
    constexpr uint32_t kOOB = 0x80000000u;
    const int srow = lane >> 3, spos = lane & 7;
    const BufRsrc rs_k = make_buf_rsrc(k + (int64_t)b * Lk * ldk + h * 64);
    const BufRsrc rs_v = make_buf_rsrc(vt + ((int64_t)(b * H + h) * 64) * ldvt);
    uint32_t k_off[2], v_off[2];
    int v_chunk[2], k_row[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rl = (wave * 2 + j) * 8 + srow;          // tile row: key (K) / d (V^T)
        const int gch = spos ^ ((rl >> 1) & 7);            // global 16-byte chunk stored at position spos
        k_row[j] = rl;
        k_off[j] = (uint32_t)((int64_t)rl * ldk * 2) + gch * 16u;
        v_off[j] = (uint32_t)((int64_t)rl * ldvt * 2) + gch * 16u;
        v_chunk[j] = gch * 8;
    }
    auto issue_tile = [&](int key0, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            buf_glds16(rs_k, key0 + k_row[j] < Lk ? k_off[j] : kOOB, (uint32_t)((int64_t)key0 * ldk * 2),
                       &KV[buf][0][(wave * 2 + j) * 8 * 64]);
            buf_glds16(rs_v, key0 + v_chunk[j] < ldvt ? v_off[j] : kOOB, (uint32_t)(key0 * 2),
                       &KV[buf][1][(wave * 2 + j) * 8 * 64]);
        }
    };

    f32x16 oacc[2], lacc;      // lacc: row sums on the matrix pipe (every register of a lane holds sum_k P[q, k])
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[0][r] = oacc[1][r] = lacc[r] = 0.f;
    float m_ref = 0.f;         // per query: the (bf16-exact) reference subtracted inside the contraction
    u16x8 qm = {0, 0, 0, 0, 0, 0, 0, 0};   // B fragment of the extra k-step: element k = 0 (lane half 0, e = 0) holds bf16(-m_ref)
    const u16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0

    const int sw_k = (pi >> 1) & 7, sw_v = (col >> 1) & 7;             // fragment-read swizzles (rows 32-aligned)
    const int nkb = (Lk + KB - 1) / KB;

    // one KV tile; CUR is a compile-time stage index so every LDS address is base + immediate
    auto tile_step = [&](int kb, auto cur_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        const int key0 = kb * KB;
        glds_wait();       // this wave's DMAs of tile kb have landed ...
        __syncthreads();   // ... everybody's have; and everybody is done reading buffer cur^1
        if (kb + 1 < nkb) issue_tile(key0 + KB, cur ^ 1);

        // ---- S^T = K Q^T  (two 32-key fragments)
        f32x16 s[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[0][r] = s[1][r] = 0.f;
        // all 8 K fragments first (8 ds_read_b128 in flight), then the 8 MFMAs: left alone, hipcc pairs every MFMA with its own
        // ds_read + s_waitcnt and exposes one LDS latency per MFMA
        u16x8 kfr[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
                kfr[ks][kf] = *(const u16x8*)&KV[cur][0][(kf * 32 + pi) * 64 + (((ks * 2 + hh) ^ sw_k) * 8)];
        PCDM_SCHED_BARRIER();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) s[kf] = mfma_32x32x16(kfr[ks][kf], qf[ks], s[kf]);
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) s[kf] = mfma_32x32x16(ones, qm, s[kf]);   // S' = S - m_ref
        // the V^T fragments of this tile are requested now, so that their LDS latency hides behind the softmax VALU work
        u16x8 vfr[4][2];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int df = 0; df < 2; ++df)
                vfr[s4][df] = *(const u16x8*)&KV[cur][1][(df * 32 + col) * 64 + (((s4 * 2 + hh) ^ sw_v) * 8)];
        PCDM_SCHED_BARRIER();
        // lane holds: s[kf][r] = score(query col, key key0 + 32kf + 16(r>>3) + 8hh + (r&7))
        if (key0 + KB > Lk) {
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + 32 * kf + 16 * (r >> 3) + 8 * hh + (r & 7) >= Lk) s[kf][r] = -1e30f;
        }
        // ---- lazy online softmax (fp32): max over this lane's 32 keys + the partner lane's 32, relative to m_ref
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const bool bump = kb == 0 || mx > thr;                 // this query's reference must move (always at the first tile)
        if (wave_any(bump)) {                                  // wave-uniform: rare after the first tiles
            const float m_new = bump ? bf2f(f2bf(m_ref + mx)) : m_ref;   // bf16-exact
            const float delta = m_new - m_ref;                 // exact: 0 for the queries that stay
            const float alpha = fast_exp2(-delta);
            m_ref = m_new;
            qm[0] = hh == 0 ? f2bf(-m_new) : (u16)0;
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kf][r] -= delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                oacc[0][r] *= alpha;
                oacc[1][r] *= alpha;
            }
            lacc[0] *= alpha;   // only register 0 is ever read back: l = l*alpha + sum_k P (added by the MFMA below)
        }
        u16x8 pf[4];
        if constexpr (ROWSUM_VALU) {
            float part[4] = {0.f, 0.f, 0.f, 0.f};   // four independent chains
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(s[kf][r]);
                    part[r & 3] += pv;
                    pf[2 * kf + (r >> 3)][r & 7] = f2bf(pv);
                }
            lacc[0] += (part[0] + part[1]) + (part[2] + part[3]);
        } else {
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r) pf[2 * kf + (r >> 3)][r & 7] = f2bf(fast_exp2(s[kf][r]));
        }
        // ---- O^T += V^T P^T ; row sums += 1^T P^T (the softmax denominator is accumulated by the matrix pipe,
        //      which has slack here, instead of 32 VALU adds per tile -- the kernel is VALU-bound at d = 64)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int df = 0; df < 2; ++df) oacc[df] = mfma_32x32x16(vfr[s4][df], pf[s4], oacc[df]);
            if constexpr (!ROWSUM_VALU) lacc = mfma_32x32x16(ones, pf[s4], lacc);
        }
    };

    issue_tile(0, 0);
    for (int kb = 0; kb < nkb; kb += 2) {
        tile_step(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < nkb) tile_step(kb + 1, std::integral_constant<int, 1>{});
    }
    // sum over all keys: the MFMA contracts over both lane halves; the VALU partial sums are combined here
    const float l_tot = ROWSUM_VALU ? lacc[0] + __shfl_xor(lacc[0], 32, 64) : lacc[0];
    const float inv = 1.0f / l_tot;
    // Output through LDS, whole rows (round 5; cdna_hip_programming.md T21): the accumulators hold 4 consecutive head dims of ONE query per
    // register quad, so direct stores are 8-byte pieces of 32 different 128-byte lines per instruction, 8 instructions per lane -- a
    // store-issue-bound tail.  The wave transposes its 32 x 64 tile through a private slice of the (now idle) K / V^T buffers (pitch 72
    // elements) and writes 4 x 16 bytes per lane: every instruction covers 8 complete 128-byte rows.
    __syncthreads();                                       // every wave is done reading the last K / V^T tile
    constexpr int OP = 72;                                 // u16 pitch of a staged row (144 B: 16-byte aligned, rows 36 banks apart)
    u16* ot = (u16*)&KV[0][0][0] + wave * (QPW * OP);
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            u16x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = f2bf(oacc[df][4 * rg + e] * inv);
            *(u16x4*)(ot + col * OP + df * 32 + 8 * rg + 4 * hh) = ov;
        }
    PCDM_WAVE_SYNC();
    {
        const int rsub = lane >> 3, ch = lane & 7;
        u16* ob = o + ((int64_t)b * Lq + q0) * ldo + h * 64 + ch * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 8 + rsub;
            if (q0 + r < Lq) *(u16x8*)(ob + (int64_t)r * ldo) = *(const u16x8*)(ot + r * OP + ch * 8);
        }
    }
}
// ---- N4: the same attention with e4m3 operands on v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales), twice the
// bf16 MFMA rate: K and V^T arrive quantised (pcdm_quantize_fp8: once per projection, every query block re-reads them), Q is scaled and
// converted once per workgroup, P is converted as it leaves the exp.  Per 64-key tile a wave issues 2 (QK^T, one per 32 keys: K = 64
// covers head_dim) + 2 (PV, one per 32 output dims: K = 64 covers the tile's keys) + 1 (row sums) fp8 MFMAs of 64 cycles and the two
// bf16 MFMAs that subtract the softmax reference -- 384 cycles against 704 of the bf16 kernel; the VALU work (one exp per score,
// maxima, conversions) is unchanged, so this pays exactly because the bf16 kernel has its matrix and vector pipes evenly loaded.
// Key order inside a tile: MFMA row i of S^T holds key e | g<<2 | h<<4 for i = e | h<<2 | g<<3, which leaves each lane half with 16
// CONSECUTIVE keys per 32-key fragment -- the 32 bytes of its P^T operand are then [frag 0: keys 16h..16h+15 | frag 1: 32+16h..],
// and the V^T operand of the same k-slots is two plain 16-byte reads.
__global__ __launch_bounds__(256, 3) void flash_attn_fp8_kernel(const u16* __restrict__ q, int64_t ldq, const uint8_t* __restrict__ k8,
                                                                int64_t ldk, const uint8_t* __restrict__ vt8, int64_t ldvt,
                                                                u16* __restrict__ o, int64_t ldo, int H, int Lq, int Lk, float c, float thr,
                                                                float out_scale, int flags) {
    // K tile [64 keys][64 B of d] and V^T tile [64 d][64 B of keys], 2 stages each; 64-byte rows, 16-byte chunks XOR-swizzled by
    // (row>>2)&3 on the DMA source and on the reads (four rows share a 256-byte bank row)
    __shared__ __attribute__((aligned(16))) uint8_t KV[2][2][KB * 64];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int qb, h, b;
    attn_block_coords(flags, qb, h, b);
    const int q0 = qb * QPB + wave * QPW;
    const int hh = lane >> 5, col = lane & 31;

    // Q operand (B of S^T = K Q^T): lane -> query col, its 32 bytes = d 32*hh .. +31, scaled to log2 units
    int qrow = q0 + col;
    const bool qvalid = qrow < Lq;
    if (!qvalid) qrow = Lq - 1;
    const u16* qp = q + ((int64_t)b * Lq + qrow) * ldq + h * 64 + hh * 32;
    // Range guard (VERDICT r3 #3e): e4m3 tops out at 448.  A query whose scaled row q c exceeds that (large to_q gains of a trained
    // checkpoint, outlier channels) is multiplied by a power of two 2^-e that brings its largest element back into range -- exact, its
    // scores are multiplied back by 2^e in fp32 behind the QK^T MFMA -- instead of being clamped (which silently changed the scores).
    // Rows in range (every row of the UNet's LayerNorm-fed to_q with unit-gain weights) take the old path: one wave-uniform branch per tile.
    u32x8 qf;
    float q_up = 1.0f;                 // 2^e of this lane's query
    {
        u16x8 raw[4];
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            raw[i] = *(const u16x8*)(qp + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f(raw[i][e]) * c));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));          // the other 32 dims of the same query
        float q_dn = 1.0f;
        if (amax > 448.f && amax < 3.0e38f) {
            // smallest e with amax 2^-e <= 448: from the exponent field of amax / 448 (rounded up)
            const uint32_t bits = __builtin_bit_cast(uint32_t, amax * (1.0f / 448.f));
            const int ex = (int)((bits >> 23) & 255) - 127 + ((bits & 0x7fffffu) ? 1 : 0);
            q_dn = __builtin_bit_cast(float, (uint32_t)((127 - ex) << 23));
            q_up = __builtin_bit_cast(float, (uint32_t)((127 + ex) << 23));
        }
        const float cq = c * q_dn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(bf2f(raw[i][e]) * cq, -448.f), 448.f);   // (only NaN / inf rows still clamp)
            qf[2 * i] = pack4_fp8(f[0], f[1], f[2], f[3]);
            qf[2 * i + 1] = pack4_fp8(f[4], f[5], f[6], f[7]);
        }
    }
    const bool q_rescaled = wave_any(q_up != 1.0f);   // wave-uniform

    // LDS-DMA staging: one K and one V^T instruction per wave per tile (1 KiB = 16 rows of 64 B each)
    constexpr uint32_t kOOB = 0x80000000u;
    const int srow = lane >> 2, spos = lane & 3;
    const int rl = wave * 16 + srow;                      // tile row: key (K) / d (V^T)
    const int gch = spos ^ ((rl >> 2) & 3);               // global 16-byte chunk stored at position spos
    const BufRsrc rs_k = make_buf_rsrc(k8 + (int64_t)b * Lk * ldk + h * 64);
    const BufRsrc rs_v = make_buf_rsrc(vt8 + ((int64_t)(b * H + h) * 64) * ldvt);
    const uint32_t k_off = (uint32_t)((int64_t)rl * ldk) + gch * 16u;
    const uint32_t v_off = (uint32_t)((int64_t)rl * ldvt) + gch * 16u;
    auto issue_tile = [&](int key0, int buf) {
        buf_glds16(rs_k, key0 + rl < Lk ? k_off : kOOB, (uint32_t)((int64_t)key0 * ldk), &KV[buf][0][wave * 1024]);
        buf_glds16(rs_v, key0 + gch * 16 < ldvt ? v_off : kOOB, (uint32_t)key0, &KV[buf][1][wave * 1024]);
    };

    f32x16 oacc[2], lacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[0][r] = oacc[1][r] = lacc[r] = 0.f;
    float m_ref = 0.f;
    u16x8 qm = {0, 0, 0, 0, 0, 0, 0, 0};
    const u16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
    const u32x8 ones8 = {0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u};   // e4m3 1.0

    const int pi = (col & 3) | ((col & 0x18) >> 1) | ((col & 4) << 2);   // MFMA row i = e | h<<2 | g<<3  ->  key e | g<<2 | h<<4
    const int sw_k = (pi >> 2) & 3, sw_v = (col >> 2) & 3;                // (fragment rows are 32-aligned: (row>>2)&3 of the row in the tile)
    const int nkb = (Lk + KB - 1) / KB;

    auto tile_step = [&](int kb, auto cur_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        const int key0 = kb * KB;
        glds_wait();
        __syncthreads();
        if (kb + 1 < nkb) issue_tile(key0 + KB, cur ^ 1);
        // ---- S^T = K Q^T: lane's 32 bytes of K row pi (+32 per fragment) = chunks 2hh, 2hh+1
        u32x8 kfr[2];
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) {
            const uint8_t* kr = &KV[cur][0][(kf * 32 + pi) * 64];
            const u32x4 lo = *(const u32x4*)(kr + (((2 * hh) ^ sw_k) * 16)), hi = *(const u32x4*)(kr + (((2 * hh + 1) ^ sw_k) * 16));
            kfr[kf] = u32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        f32x16 s[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[0][r] = s[1][r] = 0.f;
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) s[kf] = mfma_f8_32x32x64(kfr[kf], qf, s[kf]);
        if (q_rescaled) {   // (rare: see the range guard above)
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kf][r] *= q_up;
        }
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) s[kf] = mfma_32x32x16(ones, qm, s[kf]);   // S' = S - m_ref
        // V^T operands of this tile (their LDS latency hides behind the softmax): row d = 32 df + col, k-slots [16hh.. | 32+16hh..]
        u32x8 vfr[2];
#pragma unroll
        for (int df = 0; df < 2; ++df) {
            const uint8_t* vr = &KV[cur][1][(df * 32 + col) * 64];
            const u32x4 lo = *(const u32x4*)(vr + ((hh ^ sw_v) * 16)), hi = *(const u32x4*)(vr + (((2 + hh) ^ sw_v) * 16));
            vfr[df] = u32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        PCDM_SCHED_BARRIER();
        // lane holds: s[kf][r] = score(query col, key key0 + 32 kf + 16 hh + r)
        if (key0 + KB > Lk) {
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + 32 * kf + 16 * hh + r >= Lk) s[kf][r] = -1e30f;
        }
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const bool bump = kb == 0 || mx > thr;
        if (wave_any(bump)) {
            const float m_new = bump ? bf2f(f2bf(m_ref + mx)) : m_ref;
            const float delta = m_new - m_ref;
            const float alpha = fast_exp2(-delta);
            m_ref = m_new;
            qm[0] = hh == 0 ? f2bf(-m_new) : (u16)0;
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kf][r] -= delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                oacc[0][r] *= alpha;
                oacc[1][r] *= alpha;
            }
            lacc[0] *= alpha;
        }
        // P^T operand: byte 16 kf + r  <->  key 32 kf + 16 hh + r (P <= 2^thr <= 448: no clamp needed)
        u32x8 pf;
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                pf[4 * kf + j] = pack4_fp8(fast_exp2(s[kf][4 * j]), fast_exp2(s[kf][4 * j + 1]), fast_exp2(s[kf][4 * j + 2]),
                                           fast_exp2(s[kf][4 * j + 3]));
#pragma unroll
        for (int df = 0; df < 2; ++df) oacc[df] = mfma_f8_32x32x64(vfr[df], pf, oacc[df]);
        lacc = mfma_f8_32x32x64(ones8, pf, lacc);
    };

    issue_tile(0, 0);
    for (int kb = 0; kb < nkb; kb += 2) {
        tile_step(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < nkb) tile_step(kb + 1, std::integral_constant<int, 1>{});
    }
    const float inv = out_scale / lacc[0];
    if (qvalid) {
        u16* op = o + ((int64_t)b * Lq + qrow) * ldo + h * 64;
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                u16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = f2bf(oacc[df][4 * rg + e] * inv);
                *(u16x4*)(op + df * 32 + 8 * rg + 4 * hh) = ov;
            }
    }
}
}  // namespace

// XCD-aware block placement (attn_block_coords): PCDM_ATTN_XCD=0 in the environment at load time = the plain grid (A/B switch)
static int g_attn_xcd = [] { const char* e = getenv("PCDM_ATTN_XCD"); return (e && e[0] == '0') ? 0 : 1; }();

extern "C" int pcdm_flash_attn_fp8(const void* q, int64_t ldq, const void* k8, int64_t ldk, const void* vt8, int64_t ldvt, void* o,
                                   int64_t ldo, int B, int H, int Lq, int Lk, float scale, float k_descale, float v_descale,
                                   float thr_log2, pcdm_stream_t s) {
    if (!q || !k8 || !vt8 || !o || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return -1;
    if (ldq % 8 || ldk % 16 || ldvt % 16 || ldo % 4 || ldvt < Lk) return -1;
    if (!(thr_log2 >= 0.f) || thr_log2 > 8.f || !(k_descale > 0.f) || !(v_descale > 0.f)) return -1;   // P <= 2^thr must stay below 448
    if ((int64_t)Lk * ldk >= 0x7fffffffLL || (int64_t)64 * ldvt >= 0x7fffffffLL) return -2;  // 32-bit buffer offsets
    const dim3 grid((Lq + QPB - 1) / QPB, H, B);
    PCDM_LAUNCH(flash_attn_fp8_kernel, grid, dim3(256), 0, (hipStream_t)s, (const u16*)q, ldq, (const uint8_t*)k8, ldk, (const uint8_t*)vt8,
                ldvt, (u16*)o, ldo, H, Lq, Lk, scale * k_descale * 1.44269504088896341f, thr_log2, v_descale, g_attn_xcd);
    PCDM_CHECK_LAUNCH();
    return 0;
}

// The row-sum path (PCDM_ATTN_ROWSUM=valu|mfma in the environment at load time; tools/bench_attn.py).  Default since round 6: VALU -- per launch, back
// to back, the two were within 1 % of each other in every earlier round; IN the denoise step the VALU form is +0.5 % end to end in three interleaved
// same-box pairs (6.122 / 6.122 / 6.125 -> 6.160 / 6.147 / 6.150 images/s, profiles/r6_ab_attn_rowsum.json): four MFMAs fewer per key tile on a
// matrix pipe that the chip's power budget, not its issue rate, holds back in the step.
static bool g_rowsum_valu = [] { const char* e = getenv("PCDM_ATTN_ROWSUM"); return !(e && e[0] == 'm'); }();
// extra dynamic LDS per workgroup: an occupancy knob for experiments (e.g. 50000 -> 2 workgroups per CU instead of 3)
static int g_lds_pad = [] { const char* e = getenv("PCDM_ATTN_LDS_PAD"); return e ? atoi(e) : 0; }();

extern "C" int pcdm_flash_attn_thr(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                                   void* o, int64_t ldo, int B, int H, int Lq, int Lk, float scale, float thr_log2, pcdm_stream_t s) {
    if (!q || !k || !vt || !o || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return -1;
    if (ldq % 8 || ldk % 8 || ldvt % 8 || ldo % 8 || ldvt < Lk) return -1;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)o) & 15) != 0) return -1;   // (16-byte row accesses: the base addresses too -- ADVICE r5)
    if (!(thr_log2 >= 0.f) || thr_log2 > 16.f) return -1;
    if ((int64_t)Lk * ldk * 2 >= 0x7fffffffLL || (int64_t)64 * ldvt * 2 >= 0x7fffffffLL) return -2;  // 32-bit buffer offsets
    const dim3 grid((Lq + QPB - 1) / QPB, H, B);
#define PCDM_ATTN_LAUNCH(RS)                                                                                                            \
    PCDM_LAUNCH(PCDM_KERNEL_NAME(flash_attn_kernel<RS>), grid, dim3(256), g_lds_pad, (hipStream_t)s, (const u16*)q, ldq, (const u16*)k, ldk, \
                (const u16*)vt, ldvt, (u16*)o, ldo, H, Lq, Lk, scale * 1.44269504088896341f, thr_log2, g_attn_xcd)
    if (g_rowsum_valu) PCDM_ATTN_LAUNCH(true);
    else PCDM_ATTN_LAUNCH(false);
#undef PCDM_ATTN_LAUNCH
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_flash_attn(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                               void* o, int64_t ldo, int B, int H, int Lq, int Lk, float scale, pcdm_stream_t s) {
    return pcdm_flash_attn_thr(q, ldq, k, ldk, vt, ldvt, o, ldo, B, H, Lq, Lk, scale, PCDM_ATTN_DEFAULT_THR, s);
}
