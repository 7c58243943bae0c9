// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution with fused epilogues (SURVEY.md §2.1 K1-K3,K5-K7,K11).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),  fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// Design (gfx950):
//  * One workgroup = 4 waves (2x2), block tile BMxBN (128x128 or 64x64), BK = 64.  Each wave owns a
//    (BM/2)x(BN/2) sub-tile as 32x32 MFMA fragments.  The MFMA is issued "swapped" -- D = Wfrag * Xfrag^T,
//    rows = output channel n, cols = pixel/token m -- so that every lane ends up with 4 CONSECUTIVE
//    output channels of one row in each accumulator quad: the epilogue adds bias / time-embedding /
//    residual and stores 8-byte packed bf16 without any cross-lane traffic.
//  * A (activations, NHWC bf16) and W (packed [N][K], K contiguous) tiles are register-staged:
//    16-byte global loads of tile t+1 are issued before the MFMAs of tile t and written to the other
//    LDS buffer afterwards (one barrier per K-tile, cdna_hip_programming.md T14).  Register staging
//    (rather than global_load_lds) is forced by the implicit-GEMM gather: 3x3 halo / zero padding,
//    stride 2, nearest-x2 upsample folding and the two-source skip concat are all per-lane predicates.
//  * LDS rows are padded 64 -> 72 bf16 (144 B): 16 distinct rows (mod 16) x 16 B cover all 64 banks,
//    so both the ds_write_b128 staging pattern and the ds_read_b128 fragment pattern are conflict-free.
//  * Workgroup ids are remapped so each XCD (private 4 MiB L2) owns a contiguous run of M-tiles and
//    walks all N-tiles of an A tile back to back (cdna_hip_programming.md T1, bijective form).
#include "pcdm_device.h"
#include "../../include/pcdm.h"

namespace {
constexpr int BK = 64;
constexpr int LDSK = 72;  // padded row length (bf16 elements)

struct GemmArgs {
    const u16* a;
    const u16* a2;
    int64_t lda, lda2;
    int c1;
    int B, Hi, Wi, Ho, Wo, stride, upsample, cin;
    const u16* w;
    int M, N, K, Npad;
    const float* bias;
    const float* rowvec;
    int ldrv;
    int rows_per_batch;
    const u16* residual;
    int64_t ldr;
    int res_mod;
    int epilogue;
    int vt_col0;
    void* out;
    int64_t ldo;
    u16* out2;
    int64_t ldo2;
    int tiles_m, tiles_n;
};

template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 32, FN = WN / 32;
    constexpr int AR = BM / 32, BR = BN / 32;  // 16-byte chunks per thread per tile
    PCDM_DYN_SMEM(smem);
    u16* As = (u16*)smem;                    // [2][BM][LDSK]
    u16* Bs = As + 2 * BM * LDSK;            // [2][BN][LDSK]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap of the linear workgroup id
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread staging coordinates
    const int cc = t & 7;   // 16-byte chunk within the 64-wide K tile
    const int rr = t >> 3;  // row (0..31) within each 32-row slab
    int a_b[AR], a_y[AR], a_x[AR];  // conv: batch / out y / out x ; linear: a_b = row or -1
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + rr + 32 * i;
        if (m < p.M) {
            if (CONV) {
                const int hw = p.Ho * p.Wo;
                const int b = m / hw, rem = m - b * hw;
                a_b[i] = b;
                a_y[i] = rem / p.Wo;
                a_x[i] = rem - a_y[i] * p.Wo;
            } else {
                a_b[i] = m;
                a_y[i] = a_x[i] = 0;
            }
        } else {
            a_b[i] = -1;
            a_y[i] = a_x[i] = 0;
        }
    }
    const int Hv = p.Hi << p.upsample, Wv = p.Wi << p.upsample;

    u16x8 ra[AR], rb[BR];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if (CONV) {
            const int tap = k0 / p.cin, c0 = k0 - tap * p.cin;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                int iy = a_y[i] * p.stride + ky - 1, ix = a_x[i] * p.stride + kx - 1;
                const bool ok = a_b[i] >= 0 && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
                iy >>= p.upsample;
                ix >>= p.upsample;
                u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *(const u16x8*)(p.a + (((int64_t)a_b[i] * p.Hi + iy) * p.Wi + ix) * p.cin + c0 + cc * 8);
                ra[i] = v;
            }
        } else {
            const bool first = k0 < p.c1;
            const u16* src = first ? p.a : p.a2;
            const int64_t ld = first ? p.lda : p.lda2;
            const int kk = (first ? k0 : k0 - p.c1) + cc * 8;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (a_b[i] >= 0) v = *(const u16x8*)(src + (int64_t)a_b[i] * ld + kk);
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) rb[i] = *(const u16x8*)(p.w + (int64_t)(n0 + rr + 32 * i) * p.K + k0 + cc * 8);
    };
    auto store_tile = [&](int buf) {
        u16* as = As + buf * BM * LDSK;
        u16* bs = Bs + buf * BN * LDSK;
#pragma unroll
        for (int i = 0; i < AR; ++i) *(u16x8*)(as + (rr + 32 * i) * LDSK + cc * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *(u16x8*)(bs + (rr + 32 * i) * LDSK + cc * 8) = rb[i];
    };

    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = p.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const u16* as = As + cur * BM * LDSK + (wm * WM + frow) * LDSK + fk;
        const u16* bs = Bs + cur * BN * LDSK + (wn * WN + frow) * LDSK + fk;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            u16x8 xf[FM], wf[FN];
#pragma unroll
            for (int j = 0; j < FM; ++j) xf[j] = *(const u16x8*)(as + j * 32 * LDSK + ks * 16);
#pragma unroll
            for (int i = 0; i < FN; ++i) wf[i] = *(const u16x8*)(bs + i * 32 * LDSK + ks * 16);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = mfma_32x32x16(wf[i], xf[j], acc[i][j]);
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds, per (fn, fm, quad), channels n..n+3 of row m
    const int half = lane >> 5;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = m0 + wm * WM + j * 32 + (lane & 31);
        if (m >= p.M) continue;
        const int bidx = m / p.rows_per_batch;
        const int tok = m - bidx * p.rows_per_batch;
        const int64_t rrow = p.residual ? (int64_t)(m % p.res_mod) * p.ldr : 0;
        if (p.epilogue == PCDM_EPI_GEGLU) {
            if (FN == 2) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int nl = 8 * rg + 4 * half;           // 0..31 within the wave's 32 outputs
                    const int nh = n0 + wn * WN + nl;           // packed row of h
                    const int ng = nh + 32;                     // packed row of gate
                    const int no = (n0 + wn * WN) / 2 + nl;     // output channel
                    if (no >= p.N) continue;
                    const f32x4 bh = *(const f32x4*)(p.bias + nh), bg = *(const f32x4*)(p.bias + ng);
                    u16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hval = acc[0][j][4 * rg + e] + bh[e];
                        const float gval = acc[FN - 1][j][4 * rg + e] + bg[e];
                        o[e] = f2bf(hval * gelu_erf_f(gval));
                    }
                    *(u16x4*)((u16*)p.out + (int64_t)m * p.ldo + no) = o;
                }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < FN; ++i) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + wn * WN + i * 32 + 8 * rg + 4 * half;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * rg + e];
                if (p.bias) {
                    const f32x4 bv = *(const f32x4*)(p.bias + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bv[e];
                }
                if (p.rowvec) {
                    const f32x4 tv = *(const f32x4*)(p.rowvec + (int64_t)bidx * p.ldrv + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += tv[e];
                }
                if (p.residual) {
                    const u16x4 rv = *(const u16x4*)(p.residual + rrow + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
                }
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

template <int BM, int BN, bool CONV>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * LDSK * (int)sizeof(u16);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    GemmArgs g = a;
    g.tiles_m = (a.M + BM - 1) / BM;
    g.tiles_n = a.Npad / BN;
    PCDM_LAUNCH(PCDM_KERNEL_NAME(gemm_kernel<BM, BN, CONV>), dim3(g.tiles_m * g.tiles_n), dim3(256), smem, st, g);
    PCDM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" int pcdm_gemm(const pcdm_gemm_params* p, pcdm_stream_t s) {
    if (!p || !p->a || !p->w || !p->out) return -1;
    if (p->M <= 0 || p->N <= 0 || p->K <= 0 || p->K % BK || p->Npad % 64 || p->Npad < p->N || p->N % 4) return -1;
    if (p->rows_per_batch <= 0) return -1;
    GemmArgs a;
    a.a = (const u16*)p->a;
    a.a2 = (const u16*)p->a2;
    a.lda = p->lda;
    a.lda2 = p->lda2;
    a.c1 = p->a2 ? p->c1 : p->K;
    a.B = p->B; a.Hi = p->Hi; a.Wi = p->Wi; a.Ho = p->Ho; a.Wo = p->Wo;
    a.stride = p->stride; a.upsample = p->upsample; a.cin = p->cin;
    a.w = (const u16*)p->w;
    a.M = p->M; a.N = p->N; a.K = p->K; a.Npad = p->Npad;
    a.bias = p->bias;
    a.rowvec = p->rowvec;
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
    if (p->conv) {
        if (p->cin % BK || p->K != 9 * p->cin || (p->stride != 1 && p->stride != 2) || p->a2) return -1;
        if (p->upsample && p->stride != 1) return -1;
        if (p->M != p->B * p->Ho * p->Wo) return -1;
    } else {
        if (p->a2 && (p->c1 % BK || p->c1 <= 0 || p->c1 >= p->K)) return -1;
    }
    if (p->epilogue == PCDM_EPI_GEGLU && (!p->bias || p->Npad % 128 || p->N * 2 > p->Npad)) return -1;
    if (p->epilogue == PCDM_EPI_SPLIT_VT && (!p->out2 || p->vt_col0 % 4)) return -1;
    hipStream_t st = (hipStream_t)s;
    int tile = p->tile;
    if (p->epilogue == PCDM_EPI_GEGLU) tile = 1;
    if (tile == 0) {
        // 128-row tiles unless the problem is too small to fill 256 CUs with them
        const bool n128 = p->Npad % 128 == 0;
        const int64_t big = (int64_t)((p->M + 127) / 128) * (p->Npad / (n128 ? 128 : 64));
        tile = big >= 192 ? (n128 ? 1 : 3) : 2;
    }
    if (tile == 1 && p->Npad % 128) return -1;
    if (p->conv) {
        if (tile == 1) return launch_gemm<128, 128, true>(a, st);
        if (tile == 3) return launch_gemm<128, 64, true>(a, st);
        return launch_gemm<64, 64, true>(a, st);
    }
    if (tile == 1) return launch_gemm<128, 128, false>(a, st);
    if (tile == 3) return launch_gemm<128, 64, false>(a, st);
    return launch_gemm<64, 64, false>(a, st);
}
