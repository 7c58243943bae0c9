// GroupNorm(+SiLU) and LayerNorm for gfx950 -- HBM-bound kernels.
//
// Data layout: NHWC / token-major bf16, so a row of C channels is contiguous.  Every thread owns a
// fixed 8-channel octet (16-byte loads/stores, cdna_hip_programming.md G13) and walks rows, which
// keeps per-channel accumulators / affine coefficients in registers; a row of C channels is read by
// C/8 consecutive lanes (fully coalesced).  Reductions: registers -> LDS across row-lanes ->
// per-channel sums in LDS -> fp32 per-(chunk, group) partials; the apply kernel's prologue finishes them in
// fp64 (fixed order, deterministic), so a GroupNorm is two launches.  The GroupNorm input may be the virtual concat of two tensors so the
// up-block skip concat (ref stage2_inpaint_unet_2d_condition.py:792-793) is never materialised
// on the input side.
#include "pcdm_device.h"
#include "../../include/pcdm.h"

namespace {
constexpr int kGnMaxChunks = 64;
constexpr int kGnMaxC = 4096;
constexpr int kThreads = 256;

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

// part: [B][nchunk][groups][2]  per-(row-chunk, group) partial {sum, sum of squares}, fixed summation order
__global__ __launch_bounds__(kThreads) void gn_stats_kernel(const u16* __restrict__ x1, int C1,
                                                          const u16* __restrict__ x2, int C2, int HW,
                                                          int rows_per_chunk, int groups, float* __restrict__ part) {
    __shared__ float red[kThreads * 16];
    __shared__ float chs[kGnMaxC], chq[kGnMaxC];   // per-channel sums of this block's row chunk
    const int C = C1 + C2;
    const GnGeom g = gn_geom(C);
    const int t = threadIdx.x;
    const int rp = t / g.tpr, oc0 = t - rp * g.tpr;
    const bool active = rp < g.rows_par;
    const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = (r0 + rows_per_chunk < HW) ? r0 + rows_per_chunk : HW;
    for (int ocb = 0; ocb < g.noct; ocb += g.tpr) {
        const int oc = ocb + oc0;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
        if (active && oc < g.noct) {
            // 4 independent 16-byte loads in flight per thread (the loop is HBM-latency bound otherwise)
            for (int r = r0 + rp; r < r1; r += 4 * g.rows_par) {
                u16x8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = r + u * g.rows_par;
                    const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    v[u] = rr < r1 ? gn_load(x1, C1, x2, C2, (int64_t)b * HW + rr, oc * 8) : z;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = bf2f(v[u][e]);
                        s[e] += f;
                        q[e] += f * f;
                    }
            }
        }
        __syncthreads();  // previous iteration's readers are done with `red`
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[t * 16 + e] = s[e];
            red[t * 16 + 8 + e] = q[e];
        }
        __syncthreads();
        if (rp == 0 && oc < g.noct) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float ss = 0.f, qq = 0.f;
                for (int j = 0; j < g.rows_par; ++j) {
                    ss += red[(j * g.tpr + oc0) * 16 + e];
                    qq += red[(j * g.tpr + oc0) * 16 + 8 + e];
                }
                chs[oc * 8 + e] = ss;
                chq[oc * 8 + e] = qq;
            }
        }
    }
    __syncthreads();
    const int gs = C / groups;
    for (int grp = t; grp < groups; grp += kThreads) {
        float ss = 0.f, qq = 0.f;
        for (int c = grp * gs; c < (grp + 1) * gs; ++c) {
            ss += chs[c];
            qq += chq[c];
        }
        float* ps = part + (((int64_t)b * nchunk + chunk) * groups + grp) * 2;
        ps[0] = ss;
        ps[1] = qq;
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
    // finish the statistics (every block of batch row b redoes this tiny reduction: groups x nchunk x 2 floats,
    // fp64, fixed order -> bit-identical in every block and run to run; saves a launch per GroupNorm)
    for (int grp = t; grp < groups; grp += kThreads) {
        double s = 0.0, q = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) {
            const float* ps = part + (((int64_t)b * nchunk + ch) * groups + grp) * 2;
            s += (double)ps[0];
            q += (double)ps[1];
        }
        const double cnt = (double)HW * gs;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[2 * grp] = (float)mean;
        stat[2 * grp + 1] = (float)(1.0 / sqrt(var + (double)eps));
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
        for (int r = r0 + rp; r < r1; r += 4 * g.rows_par) {
            u16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * g.rows_par;
                if (rr < r1) v[u] = gn_load(x1, C1, x2, C2, (int64_t)b * HW + rr, oc * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
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

inline int gn_chunks(int HW, int rows_par) {
    int n = (HW + rows_par * 8 - 1) / (rows_par * 8);
    if (n > kGnMaxChunks) n = kGnMaxChunks;
    if (n < 1) n = 1;
    return n;
}
}  // namespace

extern "C" int64_t pcdm_groupnorm_ws_floats(int B, int C) {
    (void)C;
    return (int64_t)B * kGnMaxChunks * 256 * 2;   // [B][chunks][groups <= 256][2]
}

extern "C" int pcdm_groupnorm(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps,
                              const float* gamma, const float* beta, int fuse_silu, void* y, float* ws,
                              pcdm_stream_t s) {
    const int C = C1 + C2;
    if (!x1 || !y || !ws || B <= 0 || HW <= 0 || groups <= 0 || groups > 256) return -1;
    if (C1 % 8 || C2 % 8 || C % groups || C > kGnMaxC || (C2 > 0 && !x2)) return -1;
    hipStream_t st = (hipStream_t)s;
    const GnGeom g = gn_geom(C);
    const int nchunk = gn_chunks(HW, g.rows_par);
    const int rpc = (HW + nchunk - 1) / nchunk;
    PCDM_LAUNCH(gn_stats_kernel, dim3(nchunk, B), dim3(kThreads), 0, st, (const u16*)x1, C1, (const u16*)x2, C2, HW,
                rpc, groups, ws);
    PCDM_CHECK_LAUNCH();
    PCDM_LAUNCH(gn_apply_kernel, dim3(nchunk, B), dim3(kThreads), 0, st, (const u16*)x1, C1, (const u16*)x2, C2, HW,
                rpc, groups, eps, (const float*)ws, gamma, beta, fuse_silu, (u16*)y);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_layernorm(const void* x, void* y, int rows, int C, float eps, const float* gamma,
                              const float* beta, pcdm_stream_t s) {
    if (!x || !y || rows <= 0 || C % 8 || C > 4096 || C <= 0) return -1;
    const int rpb = kThreads / 64;
    if (C <= 1536) {
        PCDM_LAUNCH(layernorm_kernel<3>, dim3((rows + rpb - 1) / rpb), dim3(kThreads), 0, (hipStream_t)s, (const u16*)x,
                    (u16*)y, rows, C, eps, gamma, beta);
    } else {
        PCDM_LAUNCH(layernorm_kernel<8>, dim3((rows + rpb - 1) / rpb), dim3(kThreads), 0, (hipStream_t)s, (const u16*)x,
                    (u16*)y, rows, C, eps, gamma, beta);
    }
    PCDM_CHECK_LAUNCH();
    return 0;
}
