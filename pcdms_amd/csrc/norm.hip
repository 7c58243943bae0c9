// GroupNorm(+SiLU) and LayerNorm for gfx950 -- HBM-bound kernels.
//
// Data layout: NHWC / token-major bf16, so a row of C channels is contiguous.  Every thread owns a
// fixed 8-channel octet (16-byte loads/stores, cdna_hip_programming.md G13) and walks rows, which
// keeps per-channel accumulators / affine coefficients in registers; a row of C channels is read by
// C/8 consecutive lanes (fully coalesced).  Reductions: registers -> LDS across row-lanes ->
// per-channel sums in LDS -> fp32 per-(chunk, group) partials; the apply kernel's prologue finishes them in
// fp64 (fixed order, deterministic).  Three GroupNorm paths, chosen per shape by pcdm_groupnorm:
//   gn_fused_kernel    one workgroup holds the whole (image, group set) slab in registers: one launch, one pass (small feature maps)
//   gn_cluster_kernel  the slab split over S = 2 / 4 / 8 co-resident workgroups that exchange partial sums inside the launch
//                      (<= 256 workgroups = one per CU; counters in their own region of the workspace): one launch, one pass
//   gn_stats + gn_apply  two launches (level 0's widest tensors, the VAE)
// The GroupNorm input may be the virtual concat of two tensors so the up-block skip concat
// (ref stage2_inpaint_unet_2d_condition.py:792-793) is never materialised on the input side.
#include "pcdm_device.h"
#include "../../include/pcdm.h"
#include <cstdlib>

namespace {
constexpr int kGnMaxChunks = 64;
constexpr int kGnUnroll = 4;   // independent 16-byte loads in flight per thread (the passes are HBM-latency bound); measured:
                               // 128 chunks x 8 loads was 25 % slower at 64x88 (per-block reduction tail dominates)
constexpr int kGnMaxC = 4096;
constexpr int kThreads = 256;
constexpr unsigned kGnSpinLimit = 400000;   // polls of ~128 cycles (s_sleep 2 + a memory round trip): ~50 ms
constexpr int kGnClusterMaxWgs = 256;   // one workgroup per CU: every partner of a cluster is resident, whatever the dispatch order

struct GnGeom {
    int noct, tpr, rows_par;
};
__host__ __device__ inline GnGeom gn_geom(int C) {
    GnGeom g;
    g.noct = C / 8;
    g.tpr = g.noct < kThreads ? g.noct : kThreads;
    g.rows_par = kThreads / g.tpr;
    return g;
}

__device__ __forceinline__ u16x8 gn_load(const u16* x1, int C1, const u16* x2, int C2, int64_t row, int c) {
    const u16* p = (c < C1) ? (x1 + row * C1 + c) : (x2 + row * C2 + (c - C1));
    return *(const u16x8*)p;
}

// The input of the single-pass kernels: the virtual concat x1 | x2 of two bf16 tensors -- or, for x1, the tensor a split-K GEMM has NOT
// yet reduced (pcdm_gemm_params.defer_reduce): x1[row, c] = bf16( sum_s part[s][row][c] + bias[c] + rowvec[b][c] + residual[row][c] ),
// the arithmetic of gemm.hip's splitk_reduce_kernel in the same order (bit-identical to reduce-then-normalise).  A single-pass
// kernel loads every element of its slab exactly once, so it can be the reduce kernel as well: one launch, one pass over the
// partial sums, and the bf16 tensor is written (pre_out) only when something else reads it (a residual / skip connection).
struct GnSrc {
    const u16* x1;
    int C1;
    const u16* x2;
    int C2;
    const float* part;      // split-K form of x1: [S][M][ldp] fp32 partial sums (x1 is unused then)
    int S;
    int64_t slab;           // M * ldp
    int ldp;
    const float* bias;      // [>= C1] or NULL
    const float* rowvec;    // [B][ldrv] or NULL (the time-embedding projection row of the batch entry)
    const int32_t* rowvec_step;   // optional device step counter: block rowvec + *rowvec_step * rowvec_step_stride (pcdm_gemm_params)
    int64_t rowvec_step_stride;
    int rowvec_step_count;        // > 0: the counter is bounded into [0, count) on the device, *step_error = 1 when it was not (ABI 4)
    int32_t* step_error;
    int ldrv;
    const u16* residual;    // [M][ldr] or NULL
    int ldr;
    u16* pre_out;           // [M][C1] or NULL: the reduced tensor
};

__device__ __attribute__((aligned(32))) const unsigned int g_gn_zero32[8] = {0, 0, 0, 0, 0, 0, 0, 0};

template <bool SK>
__device__ __forceinline__ u16x8 gn_src_load(const GnSrc& s, int b, int64_t row, int c) {
    if (!SK || c >= s.C1) return gn_load(s.x1, s.C1, s.x2, s.C2, row, c);
    // operands first (unconditional loads; absent ones read zeros), then the slabs in their fixed order, four in flight at a time
    const f32x4 b0 = *(const f32x4*)(s.bias ? s.bias + c : (const float*)g_gn_zero32);
    const f32x4 b1 = *(const f32x4*)(s.bias ? s.bias + c + 4 : (const float*)g_gn_zero32);
    const float* rvb = s.rowvec;
    if (s.rowvec && s.rowvec_step) {
        int st = *s.rowvec_step;
        if (s.rowvec_step_count > 0 && (st < 0 || st >= s.rowvec_step_count)) {   // (as pcdm_gemm_detail::bounded_step)
            if (s.step_error) *s.step_error = 1;
            st = st < 0 ? 0 : s.rowvec_step_count - 1;
        }
        rvb = s.rowvec + (int64_t)st * s.rowvec_step_stride;
    }
    const f32x4 t0 = *(const f32x4*)(rvb ? rvb + (int64_t)b * s.ldrv + c : (const float*)g_gn_zero32);
    const f32x4 t1 = *(const f32x4*)(rvb ? rvb + (int64_t)b * s.ldrv + c + 4 : (const float*)g_gn_zero32);
    const u16x8 rv = *(const u16x8*)(s.residual ? s.residual + row * s.ldr + c : (const u16*)g_gn_zero32);
    const float* wp = s.part + row * s.ldp + c;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    int k = 0;
    for (; k + 4 <= s.S; k += 4) {
        f32x4 a[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *(const f32x4*)(wp + (k + u) * s.slab);
            d[u] = *(const f32x4*)(wp + (k + u) * s.slab + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v0 += a[u];
            v1 += d[u];
        }
    }
    for (; k < s.S; ++k) {
        v0 += *(const f32x4*)(wp + k * s.slab);
        v1 += *(const f32x4*)(wp + k * s.slab + 4);
    }
    v0 += b0;
    v1 += b1;
    v0 += t0;
    v1 += t1;
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = f2bf(v0[e] + bf2f(rv[e]));
        o[e + 4] = f2bf(v1[e] + bf2f(rv[e + 4]));
    }
    return o;
}

// the reduce alone (one thread per octet), for shapes that take the two-kernel GroupNorm path
__global__ __launch_bounds__(kThreads) void gn_reduce_kernel(const GnSrc s, int HW, int64_t rows) {
    const int noct = s.C1 / 8;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= rows * noct) return;
    const int64_t row = i / noct;
    const int c = (int)(i - row * noct) * 8;
    *(u16x8*)(s.pre_out + row * s.C1 + c) = gn_src_load<true>(s, (int)(row / HW), row, c);
}

// part: [B][nchunk][groups][2]  per-(row-chunk, group) partial {mean, M2 = sum of squares centred at that mean}, fixed merge order.
// Robust to |mean| >> std (checkpoint-like statistics; rounds 1-4 wrote {sum, sum of squares} and took E[x^2] - mean^2): a thread sums
// its rows of a channel SHIFTED by the first value it loads (x - K is of the size of the spread, so K + s/n and q - s^2/n lose
// nothing), and every merge -- the row-lanes of a channel, the channels of a group, the chunks in gn_apply -- is Chan's
// {n, mean, M2} update.
__device__ __forceinline__ void gn_chan_merge(float& n, float& mean, float& m2, float nj, float mj, float m2j) {
    if (nj <= 0.f) return;
    const float tot = n + nj, d = mj - mean, w = nj / tot;
    mean += d * w;
    m2 += m2j + d * d * n * w;
    n = tot;
}

__global__ __launch_bounds__(kThreads) void gn_stats_kernel(const u16* __restrict__ x1, int C1,
                                                          const u16* __restrict__ x2, int C2, int HW,
                                                          int rows_per_chunk, int groups, float* __restrict__ part) {
    __shared__ float red[kThreads * 16];
    __shared__ float chs[kGnMaxC], chq[kGnMaxC];   // per-channel {mean, M2} of this block's row chunk
    const int C = C1 + C2;
    const GnGeom g = gn_geom(C);
    const int t = threadIdx.x;
    const int rp = t / g.tpr, oc0 = t - rp * g.tpr;
    const bool active = rp < g.rows_par;
    const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = (r0 + rows_per_chunk < HW) ? r0 + rows_per_chunk : HW;
    // rows this thread's row-lane walks: r0 + rp, + rows_par, ... < r1
    auto lane_rows = [&](int j) { const int f = r0 + j; return f < r1 ? (r1 - f + g.rows_par - 1) / g.rows_par : 0; };
    for (int ocb = 0; ocb < g.noct; ocb += g.tpr) {
        const int oc = ocb + oc0;
        float s[8], q[8], K[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = K[e] = 0.f;
        if (active && oc < g.noct && r0 + rp < r1) {
            const u16x8 k8 = gn_load(x1, C1, x2, C2, (int64_t)b * HW + r0 + rp, oc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) K[e] = bf2f(k8[e]);
            for (int r = r0 + rp; r < r1; r += kGnUnroll * g.rows_par) {
                u16x8 v[kGnUnroll];
#pragma unroll
                for (int u = 0; u < kGnUnroll; ++u) {
                    const int rr = r + u * g.rows_par;
                    v[u] = rr < r1 ? gn_load(x1, C1, x2, C2, (int64_t)b * HW + rr, oc * 8) : k8;   // (x - K = 0 beyond the chunk)
                }
#pragma unroll
                for (int u = 0; u < kGnUnroll; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = bf2f(v[u][e]) - K[e];
                        s[e] += f;
                        q[e] += f * f;
                    }
            }
        }
        __syncthreads();  // previous iteration's readers are done with `red`
        {
            const float nt = (float)lane_rows(rp), inv = nt > 0.f ? 1.0f / nt : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[t * 16 + e] = K[e] + s[e] * inv;                 // mean of this thread's rows
                red[t * 16 + 8 + e] = q[e] - s[e] * s[e] * inv;      // M2 about it
            }
        }
        __syncthreads();
        if (rp == 0 && oc < g.noct) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float n = 0.f, mean = 0.f, m2 = 0.f;
                for (int j = 0; j < g.rows_par; ++j)
                    gn_chan_merge(n, mean, m2, (float)lane_rows(j), red[(j * g.tpr + oc0) * 16 + e], red[(j * g.tpr + oc0) * 16 + 8 + e]);
                chs[oc * 8 + e] = mean;
                chq[oc * 8 + e] = m2;
            }
        }
    }
    __syncthreads();
    const int gs = C / groups;
    const float nrow = (float)(r1 - r0);
    for (int grp = t; grp < groups; grp += kThreads) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int c = grp * gs; c < (grp + 1) * gs; ++c) gn_chan_merge(n, mean, m2, nrow, chs[c], chq[c]);
        float* ps = part + (((int64_t)b * nchunk + chunk) * groups + grp) * 2;
        ps[0] = mean;
        ps[1] = m2;
    }
}

__global__ __launch_bounds__(kThreads) void gn_apply_kernel(const u16* __restrict__ x1, int C1,
                                                          const u16* __restrict__ x2, int C2, int HW,
                                                          int rows_per_chunk, int groups, float eps,
                                                          const float* __restrict__ part,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int fuse_silu,
                                                          u16* __restrict__ y) {
    __shared__ float stat[2 * 256];  // {mean, rstd} per group of this batch row
    const int C = C1 + C2;
    const GnGeom g = gn_geom(C);
    const int t = threadIdx.x;
    const int rp = t / g.tpr, oc0 = t - rp * g.tpr;
    const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
    const int gs = C / groups;
    // finish the statistics (every block of batch row b redoes this tiny reduction: groups x nchunk x 2 floats, fixed
    // order -> bit-identical in every block and run to run; saves a launch per GroupNorm).  All 256 threads take part:
    // thread (sub, grp) sums chunks sub, sub+NSUB, ... with its loads in flight together (a serial 64-deep chain of
    // dependent L2 round trips per group cost more than the streaming pass on the small instances), then the NSUB
    // partials are combined in fp64.
    __shared__ double psum[kThreads * 3];
    {
        // Chan merge of the chunks' {mean, M2} (n_i = rows of chunk i x gs) in fp64, fixed order
        auto chunk_n = [&](int ch) {
            const int c0 = ch * rows_per_chunk;
            const int c1 = (c0 + rows_per_chunk < HW) ? c0 + rows_per_chunk : HW;
            return c1 > c0 ? (double)(c1 - c0) * gs : 0.0;
        };
        auto merge = [](double& n, double& mean, double& m2, double nj, double mj, double m2j) {
            if (nj <= 0.0) return;
            const double tot = n + nj, d = mj - mean, w = nj / tot;
            mean += d * w;
            m2 += m2j + d * d * n * w;
            n = tot;
        };
        const int nsub = kThreads / groups > 0 ? kThreads / groups : 1;   // groups <= 256
        const int sub = t / groups, grp = t - sub * groups;
        double n = 0.0, mean = 0.0, m2 = 0.0;
        if (sub < nsub) {
            for (int ch0 = sub; ch0 < nchunk; ch0 += 8 * nsub) {   // eight independent loads in flight, then their (serial) merges
                float pm[8], pq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ch = ch0 + u * nsub;
                    const float* ps = part + (((int64_t)b * nchunk + (ch < nchunk ? ch : 0)) * groups + grp) * 2;
                    pm[u] = ps[0];
                    pq[u] = ps[1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ch = ch0 + u * nsub;
                    if (ch < nchunk) merge(n, mean, m2, chunk_n(ch), (double)pm[u], (double)pq[u]);
                }
            }
        }
        psum[3 * t] = n;
        psum[3 * t + 1] = mean;
        psum[3 * t + 2] = m2;
        __syncthreads();
        if (t < groups) {
            n = mean = m2 = 0.0;
            for (int j = 0; j < nsub; ++j) merge(n, mean, m2, psum[3 * (j * groups + t)], psum[3 * (j * groups + t) + 1], psum[3 * (j * groups + t) + 2]);
            double var = n > 0.0 ? m2 / n : 0.0;
            if (var < 0.0) var = 0.0;
            stat[2 * t] = (float)mean;
            stat[2 * t + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    if (rp >= g.rows_par) return;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = (r0 + rows_per_chunk < HW) ? r0 + rows_per_chunk : HW;
    for (int oc = oc0; oc < g.noct; oc += g.tpr) {
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = oc * 8 + e;
            const int grp = c / gs;
            const float mean = stat[2 * grp], rstd = stat[2 * grp + 1];
            sc[e] = rstd * gamma[c];
            sh[e] = beta[c] - mean * sc[e];
        }
        for (int r = r0 + rp; r < r1; r += kGnUnroll * g.rows_par) {
            u16x8 v[kGnUnroll];
#pragma unroll
            for (int u = 0; u < kGnUnroll; ++u) {
                const int rr = r + u * g.rows_par;
                if (rr < r1) v[u] = gn_load(x1, C1, x2, C2, (int64_t)b * HW + rr, oc * 8);
            }
#pragma unroll
            for (int u = 0; u < kGnUnroll; ++u) {
                const int rr = r + u * g.rows_par;
                if (rr >= r1) break;
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = bf2f(v[u][e]) * sc[e] + sh[e];
                    if (fuse_silu) f = f * fast_rcp(1.0f + fast_exp2(-1.44269504088896341f * f));
                    o[e] = f2bf(f);
                }
                *(u16x8*)(y + ((int64_t)b * HW + rr) * C + oc * 8) = o;
            }
        }
    }
}

// ---- single-pass GroupNorm for instances whose per-(batch, group set) slab fits in registers (UNet levels 1-3: HW <= 1408).
// One workgroup owns `gpb` consecutive groups (gpb * gs channels = noct octets, the smallest octet-aligned run) of one
// batch image for ALL rows: the slab is read once into registers (MAXR 16-byte loads in flight per thread), mean and
// the exact centred variance are reduced in a fixed tree (deterministic), and the normalised rows are written back --
// one launch and 2 bytes moved per element instead of two launches and 3.  blockIdx -> (batch = id % B, group set = id / B)
// so that the workgroups sharing 128-byte lines of one image run on the same XCD / L2.
template <int THREADS>
__device__ __forceinline__ void gn_block_sum4(float (&x)[4], int n /* live entries (block-uniform) */, float* red /* [THREADS/64][4] */) {
    constexpr int NWV = THREADS / 64;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < n) x[k] = wave_sum(x[k]);
    __syncthreads();   // previous use of `red` is over
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wv * 4 + k] = x[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a = 0.f;
        for (int w = 0; w < NWV; ++w) a += red[w * 4 + k];
        x[k] = a;
    }
}

// Keeps the slab PACKED (bf16, 4 VGPRs per 8 values) between the three passes: without it the compiler hoists the
// bf16 -> fp32 conversions out of the passes and holds the whole slab as floats (2x the registers, occupancy 1).
#ifdef PCDM_EMU
#define GN_KEEP_PACKED(v) ((void)0)
#else
#define GN_KEEP_PACKED(v) asm volatile("" : "+v"(v))
#endif

template <int THREADS, int MAXR, bool SK>
__global__ __launch_bounds__(THREADS) void gn_fused_kernel(const GnSrc src, int B, int HW, int gs, int gpb, int noct, float eps,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int fuse_silu, u16* __restrict__ y) {
    __shared__ float red[(THREADS / 64) * 4];
    const int C = src.C1 + src.C2;
    const int b = blockIdx.x % B, gset = blockIdx.x / B;
    const int t = threadIdx.x;
    const int rows_par = THREADS / noct;
    const int rp = t / noct, oc = t - rp * noct;
    const bool active = rp < rows_par;
    const int cl = oc * 8;                       // first channel of this thread's octet, local to the group set
    const int c = gset * gpb * gs + cl;          // global channel
    int ge[8];                                   // local group of each of the 8 channels (an octet spans <= 2 groups)
#pragma unroll
    for (int e = 0; e < 8; ++e) ge[e] = (cl + e) / gs;
    u16x8 v[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int r = rp + i * rows_par;
        const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        v[i] = (active && r < HW) ? gn_src_load<SK>(src, b, (int64_t)b * HW + r, c) : z;
    }
    if (SK && src.pre_out && active && c < src.C1) {   // the reduced tensor, for the residual / skip connections that read it later
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int r = rp + i * rows_par;
            if (r < HW) *(u16x8*)(src.pre_out + ((int64_t)b * HW + r) * src.C1 + c) = v[i];
        }
    }
    // ---- mean
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXR; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += bf2f(v[i][e]);   // rows beyond HW hold zeros
    float sg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) sg[k] += (ge[e] == k) ? s[e] : 0.f;
    gn_block_sum4<THREADS>(sg, gpb, red);
    const float inv_cnt = 1.0f / ((float)HW * (float)gs);
    float mean_e[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) mean_e[e] = sg[ge[e] & 3] * inv_cnt;
    // ---- centred variance
#pragma unroll
    for (int i = 0; i < MAXR; ++i) GN_KEEP_PACKED(v[i]);
    float q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int r = rp + i * rows_par;
        if (active && r < HW) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = bf2f(v[i][e]) - mean_e[e];
                q[e] += d * d;
            }
        }
    }
    float qg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) qg[k] += (ge[e] == k) ? q[e] : 0.f;
    gn_block_sum4<THREADS>(qg, gpb, red);
    if (!active) return;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) GN_KEEP_PACKED(v[i]);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float rstd = 1.0f / sqrtf(qg[ge[e] & 3] * inv_cnt + eps);
        sc[e] = rstd * gamma[c + e];
        sh[e] = beta[c + e] - mean_e[e] * sc[e];
    }
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int r = rp + i * rows_par;
        if (r < HW) {
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = bf2f(v[i][e]) * sc[e] + sh[e];
                if (fuse_silu) f = f * fast_rcp(1.0f + fast_exp2(-1.44269504088896341f * f));
                o[e] = f2bf(f);
            }
            *(u16x8*)(y + ((int64_t)b * HW + r) * C + c) = o;
        }
    }
}

// ---- single-pass GroupNorm for slabs that do NOT fit one workgroup (UNet level 0: 5632 rows; level 1 at C = 640 with only 128
// slabs): the rows of a (batch, group set) slab are split over S workgroups that each keep their chunk in registers, publish their
// per-group {sum, sum of squares} and wait for the other S - 1 (system-scope write-through stores -> drained -> ticket; relaxed poll ->
// system-scope loads: cdna_hip_programming.md Guideline 16, the fence-free form), then every workgroup finishes the statistics itself (fixed chunk order, fp64: bit-identical
// in all of them and run to run) and normalises its rows: 2 bytes moved per element and ONE launch, where the two-kernel path moves 3
// in two launches.  The grid is at most one workgroup per CU, so all S partners are resident (no deadlock); the counters are
// self-resetting (the last workgroup to have READ the partials clears them; the next launch cannot start before this one ends).
// ws layout: [slab][S][4 groups][2] floats, then [slab][2] unsigned counters (arrived, read) -- zeroed once by the caller.
// (Round 4 measured the alternative decomposition -- a workgroup owns whole ROWS of an image with all channels, 32 workgroups per image
//  exchange every group's statistics: 640-byte accesses instead of 80-byte pieces -- at 24.2 us against this kernel's 24.0 us at level 0
//  and 18.1 against 14.1 us at level 1 (profiles/r4_bench_gn_rows_ab.txt): the access granularity is not what bounds this kernel; the
//  serial read -> exchange -> write structure is.  Dropped.)
template <int THREADS, int MAXR, bool SK>
__global__ __launch_bounds__(THREADS) void gn_cluster_kernel(const GnSrc src, int B,
                                                            int HW, int gs, int gpb, int noct, int S, int rows_per_chunk, float eps,
                                                            double inv_n /* 1 / (HW * gs) */, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int fuse_silu, u16* __restrict__ y,
                                                            float* __restrict__ ws) {
    __shared__ float red[(THREADS / 64) * 4];
    __shared__ float stat[8];                    // {mean, rstd} of the <= 4 groups of this slab
    __shared__ int stat_flag[1];                 // 1: the partners did not arrive in time (see the poll)
    const int C = src.C1 + src.C2;
    const int nslab = gridDim.x / S;
    const int slab = blockIdx.x % nslab, chunk = blockIdx.x / nslab;   // partners are nslab apart: different XCDs do not matter here
    const int b = slab % B, gset = slab / B;
    const int t = threadIdx.x;
    const int rows_par = THREADS / noct;
    const int rp = t / noct, oc = t - rp * noct;
    const bool active = rp < rows_par;
    const int cl = oc * 8;
    const int c = gset * gpb * gs + cl;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = (r0 + rows_per_chunk < HW) ? r0 + rows_per_chunk : HW;
    int ge[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ge[e] = (cl + e) / gs;
    u16x8 v[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int r = r0 + rp + i * rows_par;
        const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        v[i] = (active && r < r1) ? gn_src_load<SK>(src, b, (int64_t)b * HW + r, c) : z;
    }
    if (SK && src.pre_out && active && c < src.C1) {
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int r = r0 + rp + i * rows_par;
            if (r < r1) *(u16x8*)(src.pre_out + ((int64_t)b * HW + r) * src.C1 + c) = v[i];
        }
    }
    // ---- this chunk's per-group sum, then the sum of squares CENTRED at the chunk's own mean (exact two-pass on the registers; rows beyond
    // the chunk hold zeros and are masked out of the second pass).  The partners exchange {sum, centred M2} and every workgroup merges
    // them with Chan's formula in fp64: M2 = sum_i M2_i + n_i (mean_i - mean)^2.  (Rounds 3-4 exchanged {sum, sum of squares} and took
    // E[x^2] - mean^2: with |mean| >> std -- checkpoint-like statistics -- the fp32 partial sums of squares lose mean^2 / var of their
    // precision; tests/test_kernels.py::test_groupnorm_large_mean.)
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, qg[4] = {0.f, 0.f, 0.f, 0.f};
    {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
        for (int i = 0; i < MAXR; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += bf2f(v[i][e]);
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) sg[k] += (ge[e] == k) ? s[e] : 0.f;
    }
    gn_block_sum4<THREADS>(sg, gpb, red);
#pragma unroll
    for (int i = 0; i < MAXR; ++i) GN_KEEP_PACKED(v[i]);
    {
        const float inv_nloc = 1.0f / ((float)(r1 - r0) * (float)gs);   // (r1 > r0: the launcher's chunks are never empty)
        float mloc[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mloc[e] = sg[ge[e] & 3] * inv_nloc;
            q[e] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int r = r0 + rp + i * rows_par;
            if (active && r < r1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = bf2f(v[i][e]) - mloc[e];
                    q[e] += d * d;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) qg[k] += (ge[e] == k) ? q[e] : 0.f;
    }
    gn_block_sum4<THREADS>(qg, gpb, red);
#pragma unroll
    for (int i = 0; i < MAXR; ++i) GN_KEEP_PACKED(v[i]);
    // ---- publish, wait for the partners, finish the statistics
    float* part = ws + (int64_t)slab * S * 8;
    unsigned* cnt = (unsigned*)(ws + kGnClusterMaxWgs * 8) + slab * 2;   // fixed offset, whatever the grid (see pcdm_groupnorm_ws_floats)
    // (payload by SYSTEM-scope accesses -- write-through stores drained before the ticket, loads served by memory after the poll: no
    //  release / acquire FENCE, whose L2 write-back of the previous kernel's still-dirty output cost ~7 us per launch here.  Agent
    //  scope is not enough for the loads across XCDs: see pcdm_load_sys.)
    {   // thread t < 8 publishes entry t of {s0, q0, s1, q1, ...}: selected by compare chains -- indexing sg[] / qg[] with the lane id put
        // both arrays into a 48-byte scratch frame, i.e. a scratch round trip at the head of the exchange
        float pv = sg[0];
        pv = t == 1 ? qg[0] : pv;
        pv = t == 2 ? sg[1] : pv;
        pv = t == 3 ? qg[1] : pv;
        pv = t == 4 ? sg[2] : pv;
        pv = t == 5 ? qg[2] : pv;
        pv = t == 6 ? sg[3] : pv;
        pv = t == 7 ? qg[3] : pv;
        if (t < 8) pcdm_store_sys(part + chunk * 8 + t, pv);
    }
    pcdm_drain_vmem();
    __syncthreads();
    // BOUNDED poll: the launcher checked that the grid fits the device (occupancy x CUs), but a plain launch cannot promise co-residency
    // when something else holds CUs (another stream's long kernel, another process, a CU mask).  A workgroup whose partners have not
    // arrived within kGnSpinLimit polls (~50 ms) stops waiting and computes the statistics of the WHOLE slab itself (it re-reads the
    // other chunks' rows: slow, correct, never a hang), and counts the event in ws (pcdm_groupnorm_cluster_timeouts).
    if (t == 0) {
        pcdm_atomic_inc_agent(cnt);
        unsigned spins = 0;
        bool ok = true;
        while (pcdm_load_sys_u32(cnt) < (unsigned)S) {
            pcdm_sleep();
            if (++spins > kGnSpinLimit) { ok = false; break; }
        }
        stat_flag[0] = ok ? 0 : 1;
        if (!ok) pcdm_atomic_inc_agent((unsigned*)(ws + kGnClusterMaxWgs * 8) + kGnClusterMaxWgs * 2);
    }
    __syncthreads();
    const bool alone = stat_flag[0] != 0;   // block-uniform
    float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};
    if (alone) {   // two passes over the whole slab: sums, then squares centred at the slab mean
        for (int r = rp; r < HW && active; r += rows_par) {
            const u16x8 z = gn_src_load<SK>(src, b, (int64_t)b * HW + r, c);   // (split-K source: recomputed, same value)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf2f(z[e]);
#pragma unroll
                for (int k = 0; k < 4; ++k) fs[k] += (ge[e] == k) ? f : 0.f;
            }
        }
        gn_block_sum4<THREADS>(fs, gpb, red);
        float ma[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ma[e] = (float)((double)fs[ge[e] & 3] * inv_n);
        for (int r = rp; r < HW && active; r += rows_par) {
            const u16x8 z = gn_src_load<SK>(src, b, (int64_t)b * HW + r, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = bf2f(z[e]) - ma[e];
#pragma unroll
                for (int k = 0; k < 4; ++k) fq[k] += (ge[e] == k) ? d * d : 0.f;
            }
        }
        gn_block_sum4<THREADS>(fq, gpb, red);
    }
    if (t < gpb) {
        double mean, m2;
        if (alone) {
            mean = (double)(t == 0 ? fs[0] : t == 1 ? fs[1] : t == 2 ? fs[2] : fs[3]) * inv_n;
            m2 = (double)(t == 0 ? fq[0] : t == 1 ? fq[1] : t == 2 ? fq[2] : fq[3]);
        } else {
            // fixed chunk order, fp64: bit-identical in every workgroup of the slab and run to run.  (S <= 8; the loops are fully unrolled
            // over 8 with a guard so that ps / pq are indexed by constants and stay in registers)
            float ps[8], pq[8];
            double ss = 0.0;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                ps[ch] = pq[ch] = 0.f;
                if (ch < S) {
                    ps[ch] = pcdm_load_sys(part + ch * 8 + 2 * t);
                    pq[ch] = pcdm_load_sys(part + ch * 8 + 2 * t + 1);
                }
            }
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) ss += (double)ps[ch];
            mean = ss * inv_n;
            m2 = 0.0;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                if (ch < S) {
                    const int c0 = ch * rows_per_chunk;
                    const int c1 = (c0 + rows_per_chunk < HW) ? c0 + rows_per_chunk : HW;
                    const float n_f = (float)(c1 - c0) * (float)gs;                 // (exact: < 2^24)
                    const double d = (double)(ps[ch] / n_f) - mean;                 // chunk mean (fp32 divide: 0.5 ulp) - slab mean
                    m2 += (double)pq[ch] + (double)n_f * d * d;
                }
            }
        }
        double var = m2 * inv_n;
        if (var < 0.0) var = 0.0;
        stat[2 * t] = (float)mean;
        stat[2 * t + 1] = 1.0f / sqrtf((float)var + eps);
    }
    __syncthreads();
    if (t == 0) {   // the partials have been read: the last reader re-arms the counters for the next launch
        if (pcdm_atomic_inc_agent(cnt + 1) == (unsigned)S - 1) {
            pcdm_store_sys_u32(cnt, 0u);
            pcdm_store_sys_u32(cnt + 1, 0u);
        }
    }
    if (!active) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float mean = stat[2 * (ge[e] & 3)], rstd = stat[2 * (ge[e] & 3) + 1];
        sc[e] = rstd * gamma[c + e];
        sh[e] = beta[c + e] - mean * sc[e];
    }
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int r = r0 + rp + i * rows_par;
        if (r < r1) {
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = bf2f(v[i][e]) * sc[e] + sh[e];
                if (fuse_silu) f = f * fast_rcp(1.0f + fast_exp2(-1.44269504088896341f * f));
                o[e] = f2bf(f);
            }
            *(u16x8*)(y + ((int64_t)b * HW + r) * C + c) = o;
        }
    }
}

template <int THREADS, int MAXR>
void launch_gn_fused(hipStream_t st, const GnSrc& src, int B, int HW, int groups, int gs, int gpb,
                     int noct, float eps, const float* gamma, const float* beta, int silu, u16* y) {
    if (src.part)
        PCDM_LAUNCH(PCDM_KERNEL_NAME(gn_fused_kernel<THREADS, MAXR, true>), dim3((groups / gpb) * B), dim3(THREADS), 0, st, src, B,
                    HW, gs, gpb, noct, eps, gamma, beta, silu, y);
    else
        PCDM_LAUNCH(PCDM_KERNEL_NAME(gn_fused_kernel<THREADS, MAXR, false>), dim3((groups / gpb) * B), dim3(THREADS), 0, st, src, B,
                    HW, gs, gpb, noct, eps, gamma, beta, silu, y);
}

// One wave per row; up to NO octets per lane (NO = 3: C <= 1536, the UNet's widths; NO = 8: C <= 4096, the stage-1 prior's
// 2048); exact two-pass variance in registers.
template <int NO>
__global__ __launch_bounds__(kThreads) void layernorm_kernel(const u16* __restrict__ x, u16* __restrict__ y,
                                                           int rows, int C, float eps,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const bool valid = row < rows;  // keep whole waves alive for the shuffles
    const int noct = C / 8;
    const int64_t base = (int64_t)(valid ? row : 0) * C;
    float v[NO][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int oc = lane + 64 * j;
        if (oc < noct) {
            const u16x8 u = *(const u16x8*)(x + base + oc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[j][e] = bf2f(u[e]);
                s += v[j][e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int oc = lane + 64 * j;
        if (oc < noct) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[j][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int oc = lane + 64 * j;
        if (oc < noct) {
            const f32x4 g0 = *(const f32x4*)(gamma + oc * 8), g1 = *(const f32x4*)(gamma + oc * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(beta + oc * 8), b1 = *(const f32x4*)(beta + oc * 8 + 4);
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = f2bf((v[j][e] - mean) * rstd * g0[e] + b0[e]);
                o[e + 4] = f2bf((v[j][e + 4] - mean) * rstd * g1[e] + b1[e]);
            }
            *(u16x8*)(y + base + oc * 8) = o;
        }
    }
}

// LayerNorm with LPR lanes per row and OPL octets per lane (C = 8 * LPR * OPL): a wave normalises 64 / LPR rows at once, every
// lane is busy (the one-wave-per-row kernel above idles 24 of 64 lanes at C = 320) and has OPL independent 16-byte loads in
// flight; the LPR lanes of a row read consecutive octets (128-byte runs), reductions are log2(LPR) xor-shuffles.
template <int LPR, int OPL>
__global__ __launch_bounds__(kThreads) void layernorm_rows_kernel(const u16* __restrict__ x, u16* __restrict__ y, int rows,
                                                                float eps, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta) {
    constexpr int RPW = 64 / LPR, C = 8 * LPR * OPL;
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const int row = (blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool valid = row < rows;   // keep whole waves alive for the shuffles
    const int64_t base = (int64_t)(valid ? row : 0) * C;
    u16x8 u[OPL];
#pragma unroll
    for (int j = 0; j < OPL; ++j) u[j] = *(const u16x8*)(x + base + (j * LPR + sub) * 8);
    float v[OPL][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < OPL; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[j][e] = bf2f(u[j][e]);
            s += v[j][e];
        }
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < OPL; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[j][e] - mean;
            q += d * d;
        }
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < OPL; ++j) {
        const int c = (j * LPR + sub) * 8;
        const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
        const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = f2bf((v[j][e] - mean) * rstd * g0[e] + b0[e]);
            o[e + 4] = f2bf((v[j][e + 4] - mean) * rstd * g1[e] + b1[e]);
        }
        *(u16x8*)(y + base + c) = o;
    }
}

template <int LPR, int OPL>
void launch_layernorm_rows(hipStream_t st, const u16* x, u16* y, int rows, float eps, const float* gamma, const float* beta) {
    constexpr int rpb = (kThreads / 64) * (64 / LPR);
    PCDM_LAUNCH(PCDM_KERNEL_NAME(layernorm_rows_kernel<LPR, OPL>), dim3((rows + rpb - 1) / rpb), dim3(kThreads), 0, st, x, y, rows, eps,
                gamma, beta);
}

inline int gn_chunks(int HW, int rows_par) {
    int n = (HW + rows_par * kGnUnroll - 1) / (rows_par * kGnUnroll);
    if (n > kGnMaxChunks) n = kGnMaxChunks;
    if (n < 1) n = 1;
    return n;
}
}  // namespace

static bool gn_cluster_enabled() {
#ifdef PCDM_EMU
    return false;
#else
    // The partners of a cluster wait for each other inside the launch: every workgroup of the grid must be resident at once.  One
    // 512-thread workgroup per CU always fits, so the condition is a device with at least kGnClusterMaxWgs CUs (a full MI355X has
    // 256; a partitioned one -- CPX mode -- does not, and takes the other paths).  PCDM_GN_CLUSTER=0: A/B switch.
    // Evaluated per DEVICE (a process may drive several): the occupancy query must admit at least one 512-thread workgroup of either
    // instantiation per CU, and CUs x that >= the largest grid.  A CU mask hides CUs from the dispatcher while the attribute still
    // reports 256, so any mask in the environment disables the path.  What the query cannot see is OTHER work holding CUs at launch
    // time (a second stream, another process): the kernel's bounded poll covers that (no hang; INTEGRATION.md states the contract).
    static int state[64] = {0};   // 0 = unknown, 1 = enabled, 2 = disabled
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (state[dev] == 0) {
        bool ok = true;
        const char* e = getenv("PCDM_GN_CLUSTER");
        if (e && e[0] == '0') ok = false;
        for (const char* m : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK", "HSA_CU_MASK_SKIP_INIT"})
            if (getenv(m)) ok = false;
        int cus = 0, nb8 = 0, nb16 = 0;
        if (ok && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ok = false;
        int nb8s = 0, nb16s = 0;
        if (ok && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb8, gn_cluster_kernel<512, 8, false>, 512, 0) != hipSuccess ||
                   hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb16, gn_cluster_kernel<512, 16, false>, 512, 0) != hipSuccess ||
                   hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb8s, gn_cluster_kernel<512, 8, true>, 512, 0) != hipSuccess ||
                   hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb16s, gn_cluster_kernel<512, 16, true>, 512, 0) != hipSuccess))
            ok = false;
        nb8 = nb8 < nb8s ? nb8 : nb8s;
        nb16 = nb16 < nb16s ? nb16 : nb16s;
        if (ok && ((int64_t)cus * (nb8 < nb16 ? nb8 : nb16) < kGnClusterMaxWgs || cus < kGnClusterMaxWgs)) ok = false;
        state[dev] = ok ? 1 : 2;
    }
    return state[dev] == 1;
#endif
}

// smallest split S (2, 4, 8) of a slab over workgroups that the cluster kernel can take: <= kGnClusterMaxWgs workgroups in total and
// <= 16 rows held per thread; 0 = none
static int gn_cluster_split(int nslab, int HW, int noct) {
    const int rows_par = 512 / noct;
    for (int S = 2; S <= 8 && nslab * S <= kGnClusterMaxWgs; S *= 2) {
        const int rpc = (HW + S - 1) / S;
        if ((rpc + rows_par - 1) / rows_par <= 16) return S;
    }
    return 0;
}

static int64_t gn_ws_stats_floats(int B) { return (int64_t)B * kGnMaxChunks * 256 * 2; }
constexpr int kGnClusterFloats = kGnClusterMaxWgs * 8 + kGnClusterMaxWgs * 2 + 8;   // head of the workspace (+ the timeout counter)

extern "C" int64_t pcdm_groupnorm_ws_floats(int B, int C) {
    (void)C;
    // cluster kernel: 8 floats per workgroup, then its arrival counters (2 per slab) | [B][chunks][groups <= 256][2] partial statistics
    // of the two-kernel path.  The counters have a region of their OWN at a FIXED offset (not a function of B or of the grid: one
    // workspace serves calls of every shape): nothing else may ever write there (they are
    // zero when the workspace is allocated and every cluster launch leaves them zero) -- when they shared the area with the partial
    // statistics, a launch of another shape left float bit patterns in them, the poll fell through at once and the partners' sums were
    // read before they were written (2-4 % run-to-run differences in the full-size UNet forward).
    return kGnClusterFloats + gn_ws_stats_floats(B);
}

extern "C" int pcdm_groupnorm_cluster_timeouts(const float* ws, unsigned* count_out, pcdm_stream_t s) {
    if (!ws || !count_out) return -1;
    (void)s;
    const unsigned* src = (const unsigned*)(ws + kGnClusterMaxWgs * 8) + kGnClusterMaxWgs * 2;
#ifdef PCDM_EMU
    *count_out = *src;
    return 0;
#else
    // (a synchronous 4-byte read: diagnostics / tests only)
    return hipMemcpy(count_out, src, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1000;
#endif
}

// GroupNorm of `src` (plain or split-K form); the split-K form takes the single-pass paths as it is and the two-kernel path behind a
// reduce launch of its own (into src.pre_out, which the caller always provides)
static int gn_launch(GnSrc src, int B, int HW, int groups, float eps, const float* gamma, const float* beta, int fuse_silu, void* y,
                     float* ws, hipStream_t st, u16* reduce_buf = nullptr /* split-K form: [M][C1] for the two-kernel path */) {
    const int C1 = src.C1, C2 = src.C2, C = C1 + C2;
    {   // single-pass path: the (batch, group set) slab in registers
        const int gs = C / groups;
        int gpb = 1;
        while (gpb <= 4 && (gpb * gs) % 8) ++gpb;
        const int noct = gpb * gs / 8;
        // slab per workgroup = HW * noct * 16 bytes; beyond ~352 KiB (level 0: 5632 rows) too few, too long workgroups
        static const int64_t max_slab = [] {
            const char* e = getenv("PCDM_GN_FUSED_MAX_KB");   // tuning knob (tools/bench_ops.py); 0 disables the fused path
            return (int64_t)(e ? atoi(e) : 352) * 1024;
        }();
        const bool few_slabs = (groups / gpb) * B <= 128 && HW >= 1024;   // <= 128 long workgroups: the cluster kernel below fills the chip
        // (only when the cluster kernel can actually take the shape: otherwise the single-pass path stays the better one)
        const bool to_cluster = few_slabs && gpb <= 4 && groups % gpb == 0 && noct <= 64 && gn_cluster_enabled() &&
                                gn_cluster_split((groups / gpb) * B, HW, noct) > 0;
        if (gpb <= 4 && groups % gpb == 0 && noct <= 64 && (int64_t)HW * noct * 16 <= max_slab && !to_cluster) {
            // rows per thread at 256 / 512 / 1024 threads, at most 8 (all loads of the slab in flight at once, <= 128 KiB)
            auto need = [&](int th) { return (HW + th / noct - 1) / (th / noct); };
            bool done = true;
#define GN_FUSED(TH, MR) launch_gn_fused<TH, MR>(st, src, B, HW, groups, gs, gpb, noct, eps, gamma, beta, fuse_silu, (u16*)y)
            if (need(256) <= 8) GN_FUSED(256, 8);
            else if (need(512) <= 8) GN_FUSED(512, 8);
            else if (need(1024) <= 8) GN_FUSED(1024, 8);
#undef GN_FUSED
            else done = false;
            if (done) {
                PCDM_CHECK_LAUNCH();
                return 0;
            }
        }
    }
#ifndef PCDM_EMU   // (the lane emulator runs workgroups one after the other: workgroups that wait for each other cannot run there;
                   //  the cluster kernel is covered by the -m gpu tests only)
    {   // cluster path: a slab split over S workgroups, at most one workgroup per CU in total (all partners resident)
        const int gs = C / groups;
        int gpb = 1;
        while (gpb <= 4 && (gpb * gs) % 8) ++gpb;
        const int noct = gpb * gs / 8;
        if (gn_cluster_enabled() && gpb <= 4 && groups % gpb == 0 && noct <= 64) {
            const int nslab = (groups / gpb) * B;
            const int rows_par = 512 / noct;   // 512 threads: 256 registers per lane, no spills with 16 rows held per thread
            if (const int S = gn_cluster_split(nslab, HW, noct)) {
                const int rpc = (HW + S - 1) / S;
                const int need = (rpc + rows_par - 1) / rows_par;
                float* cws = ws;   // cluster area at the head of the workspace: [kGnClusterMaxWgs][8] partials, then the counters
                const double inv_n = 1.0 / ((double)HW * gs);
#define GN_CLUSTER(MR, SK_)                                                                                                              \
    PCDM_LAUNCH(PCDM_KERNEL_NAME(gn_cluster_kernel<512, MR, SK_>), dim3(nslab * S), dim3(512), 0, st, src, B, HW, gs, gpb, noct, S, rpc, eps, \
                inv_n, gamma, beta, fuse_silu, (u16*)y, cws)
                if (need <= 8) {
                    if (src.part) GN_CLUSTER(8, true);
                    else GN_CLUSTER(8, false);
                } else {
                    if (src.part) GN_CLUSTER(16, true);
                    else GN_CLUSTER(16, false);
                }
#undef GN_CLUSTER
                PCDM_CHECK_LAUNCH();
                return 0;
            }
        }
    }
#endif
    if (src.part) {   // two-kernel path: reduce first (one launch more, as before the fusion), then the plain tensor
        const int64_t rows = (int64_t)B * HW, n = rows * (C1 / 8);
        src.pre_out = reduce_buf;
        PCDM_LAUNCH(gn_reduce_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, src, HW, rows);
        PCDM_CHECK_LAUNCH();
        src.x1 = src.pre_out;
        src.part = nullptr;
    }
    const GnGeom g = gn_geom(C);
    const int nchunk = gn_chunks(HW, g.rows_par);
    const int rpc = (HW + nchunk - 1) / nchunk;
    ws += kGnClusterFloats;   // (the head belongs to the cluster kernel)
    PCDM_LAUNCH(gn_stats_kernel, dim3(nchunk, B), dim3(kThreads), 0, st, src.x1, C1, src.x2, C2, HW, rpc, groups, ws);
    PCDM_CHECK_LAUNCH();
    PCDM_LAUNCH(gn_apply_kernel, dim3(nchunk, B), dim3(kThreads), 0, st, src.x1, C1, src.x2, C2, HW, rpc, groups, eps, (const float*)ws, gamma,
                beta, fuse_silu, (u16*)y);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_groupnorm(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps,
                              const float* gamma, const float* beta, int fuse_silu, void* y, float* ws,
                              pcdm_stream_t s) {
    const int C = C1 + C2;
    if (!x1 || !y || !ws || B <= 0 || HW <= 0 || groups <= 0 || groups > 256) return -1;
    if (C1 % 8 || C2 % 8 || C % groups || C > kGnMaxC || (C2 > 0 && !x2)) return -1;
    GnSrc src{};
    src.x1 = (const u16*)x1; src.C1 = C1; src.x2 = (const u16*)x2; src.C2 = C2;
    return gn_launch(src, B, HW, groups, eps, gamma, beta, fuse_silu, y, ws, (hipStream_t)s);
}

extern "C" int pcdm_groupnorm_splitk(const pcdm_gn_splitk_src* p, const void* x2, int C2, int B, int HW, int groups, float eps,
                                     const float* gamma, const float* beta, int fuse_silu, void* y, float* ws, pcdm_stream_t s) {
    if (!p || !p->part || !p->pre_out || !y || !ws || B <= 0 || HW <= 0 || groups <= 0 || groups > 256) return -1;
    const int C1 = p->N, C = C1 + C2;
    if (C1 <= 0 || C1 % 8 || C2 % 8 || C % groups || C > kGnMaxC || (C2 > 0 && !x2)) return -1;
    if (p->split_k < 2 || p->split_k > 64 || p->Npad < C1 || p->Npad % 8 || p->M != B * HW) return -1;
    if ((p->rowvec && (p->ldrv % 4 || ((uintptr_t)p->rowvec & 15) || (p->rowvec_step && p->rowvec_step_stride % 4))) || ((uintptr_t)p->bias & 15) ||
        (p->residual && (p->ldr < C1 || p->ldr % 8))) return -1;
    GnSrc src{};
    src.x1 = (const u16*)p->pre_out; src.C1 = C1; src.x2 = (const u16*)x2; src.C2 = C2;
    src.part = p->part; src.S = p->split_k; src.ldp = p->Npad; src.slab = (int64_t)p->M * p->Npad;
    src.bias = p->bias; src.rowvec = p->rowvec; src.ldrv = (int)p->ldrv;
    src.rowvec_step = p->rowvec ? p->rowvec_step : nullptr; src.rowvec_step_stride = p->rowvec_step_stride; src.residual = (const u16*)p->residual; src.ldr = (int)p->ldr;
    src.rowvec_step_count = src.rowvec_step ? p->rowvec_step_count : 0; src.step_error = src.rowvec_step ? p->step_error : nullptr;
    if (src.rowvec_step_count < 0 || ((uintptr_t)src.step_error & 3)) return -1;
    src.pre_out = p->store_pre ? (u16*)p->pre_out : nullptr;   // (the two-kernel path writes the buffer whatever the flag)
    return gn_launch(src, B, HW, groups, eps, gamma, beta, fuse_silu, y, ws, (hipStream_t)s, (u16*)p->pre_out);
}

extern "C" int pcdm_layernorm(const void* x, void* y, int rows, int C, float eps, const float* gamma,
                              const float* beta, pcdm_stream_t s) {
    if (!x || !y || rows <= 0 || C % 8 || C > 4096 || C <= 0) return -1;
    const int rpb = kThreads / 64;
    hipStream_t st = (hipStream_t)s;
    const u16* xi = (const u16*)x;
    u16* yo = (u16*)y;
    switch (C) {   // widths of the UNet (320 / 640 / 1280), DINOv2 (1536), the prior (2048), ImageProjModel_p (768)
        case 320: launch_layernorm_rows<8, 5>(st, xi, yo, rows, eps, gamma, beta); break;
        case 640: launch_layernorm_rows<16, 5>(st, xi, yo, rows, eps, gamma, beta); break;
        case 1280: launch_layernorm_rows<32, 5>(st, xi, yo, rows, eps, gamma, beta); break;
        case 768: launch_layernorm_rows<32, 3>(st, xi, yo, rows, eps, gamma, beta); break;
        case 1536: launch_layernorm_rows<64, 3>(st, xi, yo, rows, eps, gamma, beta); break;
        case 2048: launch_layernorm_rows<64, 4>(st, xi, yo, rows, eps, gamma, beta); break;
        default:
            if (C <= 1536) {
                PCDM_LAUNCH(layernorm_kernel<3>, dim3((rows + rpb - 1) / rpb), dim3(kThreads), 0, st, xi, yo, rows, C, eps, gamma, beta);
            } else {
                PCDM_LAUNCH(layernorm_kernel<8>, dim3((rows + rpb - 1) / rpb), dim3(kThreads), 0, st, xi, yo, rows, C, eps, gamma, beta);
            }
    }
    PCDM_CHECK_LAUNCH();
    return 0;
}
