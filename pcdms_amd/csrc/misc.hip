// Small / elementwise kernels of the stage-2 denoise step (SURVEY.md §2.1 K12, K13 and the
// input assembly of stage2_inpaint_pipeline.py:499-501).  All HBM- or latency-bound; 16-byte
// vector accesses where the layout allows, device-resident step index so one captured hipGraph
// replays for every timestep.
#include "pcdm_device.h"
#include "../../include/pcdm.h"

namespace {
// out[b, j]: diffusers Timesteps (flip_sin_to_cos => [cos | sin])
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t_dev, const int32_t* __restrict__ step_dev,
                                          float* __restrict__ out, int B, int dim, int flip, float shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int j = i % dim, half = dim / 2;
    const float tv = (float)t_dev[step_dev ? *step_dev : 0];
    const int kk = j < half ? j : j - half;
    const float f = expf(-9.210340371976184f * (float)kk / ((float)half - shift));
    const float a = tv * f;
    const bool is_cos = flip ? (j < half) : (j >= half);
    out[i] = is_cos ? cosf(a) : sinf(a);
}

// the same for a table of timesteps: row i <- t_dev[i]
__global__ void timestep_embedding_rows_kernel(const int64_t* __restrict__ t_dev, float* __restrict__ out, int n, int dim, int flip, float shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dim) return;
    const int j = i % dim, half = dim / 2;
    const float tv = (float)t_dev[i / dim];
    const int kk = j < half ? j : j - half;
    const float f = expf(-9.210340371976184f * (float)kk / ((float)half - shift));
    const float a = tv * f;
    const bool is_cos = flip ? (j < half) : (j >= half);
    out[i] = is_cos ? cosf(a) : sinf(a);
}

// out[(i * B + b) * D + d] = bf16( silu( emb_t[i * D + d] + cls[b * D + d] ) )   (cls may be NULL)
__global__ void time_class_combine_kernel(const float* __restrict__ emb_t, const float* __restrict__ cls, u16* __restrict__ out, int n, int B, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * B * D) return;
    const int d = (int)(i % D);
    const int64_t r = i / D;
    const int b = (int)(r % B), st = (int)(r / B);
    float v = emb_t[(int64_t)st * D + d];
    if (cls) v += cls[(int64_t)b * D + d];
    out[i] = f2bf(silu_f(v));
}

// y[b, n] = act_out( sum_k act_in(x[b,k]) W[n,k] + bias[n] ) + add[b,n]; one wave per n.
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const u16* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ add, float* __restrict__ y,
                                                           int B, int K, int N, int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool valid = n < N;
    const u16* wr = w + (int64_t)(valid ? n : 0) * K;
    for (int b0 = 0; b0 < B; b0 += 8) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int k = lane * 8; k < K; k += 64 * 8) {
            const u16x8 wv = *(const u16x8*)(wr + k);
            float wf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[e] = bf2f(wv[e]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (b0 + i < B) {
                    const float* xr = x + (int64_t)(b0 + i) * K + k;
                    const f32x4 x0 = *(const f32x4*)xr, x1 = *(const f32x4*)(xr + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a0 = x0[e], a1 = x1[e];
                        if (act_in) {
                            a0 = silu_f(a0);
                            a1 = silu_f(a1);
                        }
                        acc[i] += a0 * wf[e] + a1 * wf[e + 4];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s = wave_sum(acc[i]);
            if (lane == 0 && valid && b0 + i < B) {
                float v = s + (bias ? bias[n] : 0.f);
                if (act_out == 1) v = silu_f(v);
                if (add) v += add[(int64_t)(b0 + i) * N + n];
                if (act_out == 2) v = silu_f(v);   // activation of the SUM (emb = time + class, consumed as silu(emb))
                y[(int64_t)(b0 + i) * N + n] = v;
            }
        }
    }
}

__global__ void assemble_input_kernel(const float* __restrict__ latents, int N, int rep,
                                      const float* __restrict__ mask, int mask_b, const float* __restrict__ masked,
                                      int masked_b, u16* __restrict__ out, int HW, int cpad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, pixel, octet)
    const int noct = cpad / 8;
    const int64_t total = (int64_t)N * rep * HW * noct;
    if (i >= total) return;
    const int oc = (int)(i % noct);
    const int64_t bp = i / noct;
    const int pix = (int)(bp % HW), b = (int)(bp / HW);
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = oc * 8 + e;
        float v = 0.f;
        const int c0 = mask ? 5 : 4;   // no mask channel: [latents | masked] (the stage-3 refinement input)
        if (c < 4) v = latents[((int64_t)(b % N) * 4 + c) * HW + pix];
        else if (mask && c == 4) v = mask[(int64_t)(mask_b == 1 ? 0 : b) * HW + pix];
        else if (c >= c0 && c < c0 + 4) v = masked[((int64_t)(masked_b == 1 ? 0 : b) * 4 + (c - c0)) * HW + pix];
        o[e] = f2bf(v);
    }
    *(u16x8*)(out + i * 8) = o;
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, u16* __restrict__ y, int C, int Cpad, int HW,
                                    int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, pixel, octet)
    if (i >= total) return;
    const int noct = Cpad / 8;
    const int oc = (int)(i % noct);
    const int64_t bp = i / noct;
    const int pix = (int)(bp % HW);
    const int64_t b = bp / HW;
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = oc * 8 + e;
        o[e] = c < C ? f2bf(x[(b * C + c) * HW + pix]) : (u16)0;
    }
    *(u16x8*)(y + i * 8) = o;
}

__global__ void nhwc_to_nchw_kernel(const u16* __restrict__ x, float* __restrict__ y, int C, int HW, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, c, pix)
    if (i >= total) return;
    const int pix = (int)(i % HW);
    const int64_t bc = i / HW;
    const int c = (int)(bc % C);
    const int64_t b = bc / C;
    y[i] = bf2f(x[(b * HW + pix) * C + c]);
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, u16* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = f2bf(x[i]);
}

__global__ void cfg_step_kernel(const float* __restrict__ eps, int cfg, float g, const float* __restrict__ x,
                                const float* __restrict__ noise, float* __restrict__ x_prev,
                                float* __restrict__ eps_out, const float* __restrict__ coef,
                                const int32_t* __restrict__ step_dev, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* cf = coef + 4 * (step_dev ? *step_dev : 0);
    float e = eps[i];
    if (cfg) e = e + g * (eps[n + i] - e);
    if (eps_out) eps_out[i] = e;
    if (x_prev) {
        float v = cf[0] * x[i] + cf[1] * e;
        if (noise) v += cf[2] * noise[i];
        x_prev[i] = v;
    }
}

// UniPCMultistepScheduler.step (order <= 2, predict_x0, bh1 / bh2) as ONE elementwise kernel on static
// state slots, so the whole UniPC update is graph-replayable: the step's twelve host-computed scalars come from a DEVICE table
// row selected by the device step counter.
//   m_t   = c[0] x + c[1] eps                                   (x0-prediction, convert_model_output)
//   x_c   = c[2] != 0 ? c[3] last + c[4] m1 + c[5] m2 + c[6] m_t : x     (corrector; m1 = newest stored output, m2 the one before)
//   x'    = c[7] x_c + c[8] m_t + c[9] m1                        (predictor, after the history shift m_t -> m1 -> m2)
// and the state advances in place: x <- x', m2 <- m1, m1 <- m_t, last <- x_c.  Every element is owned by one thread.
__global__ void unipc_step_kernel(const float* __restrict__ eps, int cfg, float g, float* __restrict__ x, float* __restrict__ m1,
                                  float* __restrict__ m2, float* __restrict__ last, const float* __restrict__ coef,
                                  const int32_t* __restrict__ step_dev, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* c = coef + 12 * (step_dev ? *step_dev : 0);
    float e = eps[i];
    if (cfg) e = e + g * (eps[n + i] - e);
    const float xi = x[i], a1 = m1[i], a2 = m2[i];
    const float mt = c[0] * xi + c[1] * e;
    float xc = xi;
    if (c[2] != 0.f) xc = c[3] * last[i] + c[4] * a1 + c[5] * a2 + c[6] * mt;
    x[i] = c[7] * xc + c[8] * mt + c[9] * a1;
    m2[i] = a1;
    m1[i] = mt;
    last[i] = xc;
}

// UnCLIPScheduler.step on a [N, n/N] vector (stage-1 prior): guided prediction -> x0 -> clip -> posterior mean (+ noise),
// then an optional affine read-out (post_process_latents).  c = {p_x, p_e, clip, c_x0, c_x, c_noise, out_scale, out_shift}.
struct UnclipArgs { float c[8]; };
__global__ void unclip_step_kernel(const float* __restrict__ pred, int cfg, float g, const float* __restrict__ x,
                                   const float* __restrict__ noise, float* __restrict__ x_prev, UnclipArgs a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float e = pred[i];
    if (cfg) e = e + g * (pred[n + i] - e);
    const float xi = x[i];
    float x0 = a.c[0] * xi + a.c[1] * e;
    if (a.c[2] > 0.f) x0 = fminf(fmaxf(x0, -a.c[2]), a.c[2]);
    float v = a.c[3] * x0 + a.c[4] * xi;
    if (noise) v += a.c[5] * noise[i];
    x_prev[i] = v * a.c[6] + a.c[7];
}

// the same with the step's eight coefficients and its noise slab taken from DEVICE tables at *step_dev (hipGraph-replayable stage-1 loop)
__global__ void unclip_step_dev_kernel(const float* __restrict__ pred, int cfg, float g, float* __restrict__ x,
                                       const float* __restrict__ noise_all, const float* __restrict__ coef,
                                       const int32_t* __restrict__ step_dev, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int st = *step_dev;
    const float* c = coef + 8 * st;
    float e = pred[i];
    if (cfg) e = e + g * (pred[n + i] - e);
    const float xi = x[i];
    float x0 = c[0] * xi + c[1] * e;
    if (c[2] > 0.f) x0 = fminf(fmaxf(x0, -c[2]), c[2]);
    float v = c[3] * x0 + c[4] * xi;
    if (noise_all && c[5] != 0.f) v += c[5] * noise_all[(int64_t)st * n + i];
    x[i] = v * c[6] + c[7];
}

struct LinArgs {
    const float* x[6];
    float c[6];
};
__global__ void lincomb_kernel(float* __restrict__ y, int nin, LinArgs a, const float* __restrict__ cdev, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int j = 0; j < nin; ++j) v += (cdev ? cdev[j] : a.c[j]) * a.x[j][i];
    y[i] = v;
}

// rescale_noise_cfg (ref stage2_inpaint_pipeline.py:52-63): per-sample unbiased std of the guided eps and of the
// conditional eps over all C*H*W elements; out = gr * cfg * (std_text / std_cfg) + (1 - gr) * cfg.
// One workgroup per sample; fp64 block reduction (wave64 shuffles + LDS).
__global__ __launch_bounds__(1024) void rescale_cfg_kernel(const float* __restrict__ cfg_eps,
                                                           const float* __restrict__ text_eps, float* __restrict__ out,
                                                           int64_t n, float gr) {
    __shared__ double red[16][4];
    __shared__ float factor;
    const float* a = cfg_eps + (int64_t)blockIdx.x * n;
    const float* b = text_eps + (int64_t)blockIdx.x * n;
    double s[4] = {0, 0, 0, 0};
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = a[i], y = b[i];
        s[0] += x; s[1] += x * x; s[2] += y; s[3] += y * y;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s[k] += __shfl_xor(s[k], m, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int k = 0; k < 4; ++k) red[wave][k] = s[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4] = {0, 0, 0, 0};
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
            for (int k = 0; k < 4; ++k) t[k] += red[w][k];
        const double var_c = (t[1] - t[0] * t[0] / (double)n) / (double)(n - 1);
        const double var_t = (t[3] - t[2] * t[2] / (double)n) / (double)(n - 1);
        factor = (float)sqrt(var_t / var_c);
    }
    __syncthreads();
    const float f = gr * factor + (1.0f - gr);
    float* o = out + (int64_t)blockIdx.x * n;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) o[i] = a[i] * f;
}

// p[r, :] = softmax(scale * s[r, :]) as bf16; one workgroup (256 threads) per row, the row lives in registers
// (cols <= 8192), fp32 max / exp / sum with wave64 shuffles + LDS across the 4 waves.  Used by the VAE's single-head
// d=512 attention (two MFMA GEMMs around it), SURVEY.md §8f N1.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, u16* __restrict__ p, int cols,
                                                           int64_t ld_s, int64_t ld_p, float scale_log2e) {
    __shared__ float red[4];
    const float* sr = s + (int64_t)blockIdx.x * ld_s;
    u16* pr = p + (int64_t)blockIdx.x * ld_p;
    const int t = threadIdx.x;
    float v[32];
    float mx = -1e30f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = t + 256 * i;
        v[i] = c < cols ? sr[c] * scale_log2e : -1e30f;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        v[i] = fast_exp2(v[i] - mx);
        sum += (t + 256 * i < cols) ? v[i] : 0.f;
    }
    sum = wave_sum(sum);
    if ((t & 63) == 0) red[t >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = t + 256 * i;
        if (c < cols) pr[c] = f2bf(v[i] * inv);
    }
}

// DiagonalGaussianDistribution.sample() * scaling_factor (ref stage2_inpaint_pipeline.py:443-444):
// moments fp32 [B, 2*zc, HW] (mean | logvar); out = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale
__global__ void gaussian_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise,
                                       float* __restrict__ out, int zc, int HW, float scale, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over (b, c, pix)
    if (i >= total) return;
    const int64_t chw = (int64_t)zc * HW;
    const int64_t b = i / chw, r = i - b * chw;
    const float mean = mom[b * 2 * chw + r];
    float lv = mom[b * 2 * chw + chw + r];
    lv = fminf(fmaxf(lv, -30.0f), 20.0f);
    const float z = noise ? noise[i] : 0.f;
    out[i] = (mean + __expf(0.5f * lv) * z) * scale;
}

// VaeImageProcessor.postprocess: (x/2+0.5).clamp(0,1) -> NHWC -> round(255 x) -> uint8.  x fp32 [B, cstride>=3, HW] (NCHW)
__global__ void image_to_uint8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int cstride, int HW,
                                      int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over (b, pix)
    if (i >= total) return;
    const int64_t b = i / HW, pix = i - b * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = x[(b * cstride + c) * HW + pix] * 0.5f + 0.5f;
        v = fminf(fmaxf(v, 0.f), 1.f);
        out[i * 3 + c] = (uint8_t)rintf(v * 255.0f);
    }
}

// out NHWC [B, 2H, 2W, C] <- in [B H W, 4 C] (phase 2a + b major): the pixel shuffle behind the phase-decomposed Upsample2D convolution.
// One thread per 16 bytes of output: consecutive threads walk the channels of one output pixel, then the next pixel of the output row.
__global__ void pixel_shuffle2_kernel(const u16* __restrict__ in, u16* __restrict__ out, int H, int W, int C8, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    const int64_t px = i / C8;                  // output pixel index (b, Y, X)
    const int W2 = 2 * W;
    const int X = (int)(px % W2);
    const int64_t bY = px / W2;                 // b * 2H + Y
    const int Y = (int)(bY % (2 * H));
    const int64_t b = bY / (2 * H);
    const int64_t src = (((b * H + (Y >> 1)) * W + (X >> 1)) * 4 + ((Y & 1) * 2 + (X & 1))) * C8 + c8;
    ((u32x4*)out)[i] = ((const u32x4*)in)[src];
}

__global__ void advance_step_kernel(int32_t* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1;
}

inline dim3 grid1d(int64_t n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }
}  // namespace

extern "C" int pcdm_version(void) { return PCDM_ABI_VERSION; }   // 5: pcdm_gemm_params starts with struct_size, ends with a3 / lda3 (include/pcdm.h)
extern "C" int pcdm_is_emulator(void) {
#ifdef PCDM_EMU
    return 1;
#else
    return 0;
#endif
}

extern "C" int pcdm_timestep_embedding(const int64_t* t_dev, const int32_t* step_dev, float* out, int B, int dim,
                                       int flip_sin_to_cos, float shift, pcdm_stream_t s) {
    if (!t_dev || !out || B <= 0 || dim <= 0 || dim % 2) return -1;
    PCDM_LAUNCH(timestep_embedding_kernel, grid1d((int64_t)B * dim, 256), dim3(256), 0, (hipStream_t)s, t_dev, step_dev,
                out, B, dim, flip_sin_to_cos, shift);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_timestep_embedding_rows(const int64_t* t_dev, int n, float* out, int dim, int flip_sin_to_cos, float shift, pcdm_stream_t s) {
    if (!t_dev || !out || n <= 0 || dim <= 0 || dim % 2) return -1;
    PCDM_LAUNCH(timestep_embedding_rows_kernel, grid1d((int64_t)n * dim, 256), dim3(256), 0, (hipStream_t)s, t_dev, out, n, dim, flip_sin_to_cos, shift);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_time_class_combine(const float* emb_t, const float* cls, void* out_bf16, int n, int B, int D, pcdm_stream_t s) {
    if (!emb_t || !out_bf16 || n <= 0 || B <= 0 || D <= 0) return -1;
    PCDM_LAUNCH(time_class_combine_kernel, grid1d((int64_t)n * B * D, 256), dim3(256), 0, (hipStream_t)s, emb_t, cls, (u16*)out_bf16, n, B, D);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_small_linear(const float* x, const void* w, const float* bias, const float* add, float* y, int B,
                                 int K, int N, int act_in, int act_out, pcdm_stream_t s) {
    if (!x || !w || !y || B <= 0 || B > 32 || K <= 0 || K % 8 || N <= 0) return -1;
    PCDM_LAUNCH(small_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)s, x, (const u16*)w, bias, add, y,
                B, K, N, act_in, act_out);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_assemble_input(const float* latents, int N, int rep, const float* mask, int mask_b,
                                   const float* masked, int masked_b, void* out, int h, int w, int cpad,
                                   pcdm_stream_t s) {
    if (!latents || !masked || !out || N <= 0 || rep <= 0 || cpad % 8 || cpad < 16) return -1;   // mask == NULL: [latents | masked]
    const int64_t total = (int64_t)N * rep * h * w * (cpad / 8);
    PCDM_LAUNCH(assemble_input_kernel, grid1d(total, 256), dim3(256), 0, (hipStream_t)s, latents, N, rep, mask, mask_b,
                masked, masked_b, (u16*)out, h * w, cpad);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_nchw_f32_to_nhwc_bf16(const float* x, void* y, int B, int C, int Cpad, int HW, pcdm_stream_t s) {
    if (!x || !y || Cpad % 8 || Cpad < C) return -1;
    const int64_t total = (int64_t)B * HW * (Cpad / 8);
    PCDM_LAUNCH(nchw_to_nhwc_kernel, grid1d(total, 256), dim3(256), 0, (hipStream_t)s, x, (u16*)y, C, Cpad, HW, total);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_nhwc_bf16_to_nchw_f32(const void* x, float* y, int B, int C, int HW, pcdm_stream_t s) {
    if (!x || !y) return -1;
    const int64_t total = (int64_t)B * HW * C;
    PCDM_LAUNCH(nhwc_to_nchw_kernel, grid1d(total, 256), dim3(256), 0, (hipStream_t)s, (const u16*)x, y, C, HW, total);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_f32_to_bf16(const float* x, void* y, int64_t n, pcdm_stream_t s) {
    if (!x || !y || n <= 0) return -1;
    PCDM_LAUNCH(f32_to_bf16_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)s, x, (u16*)y, n);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_cfg_step(const float* eps, int cfg, float g, const float* x, const float* noise, float* x_prev,
                             float* eps_out, const float* coef, const int32_t* step_dev, int64_t n, pcdm_stream_t s) {
    if (!eps || n <= 0 || (x_prev && (!x || !coef))) return -1;
    PCDM_LAUNCH(cfg_step_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)s, eps, cfg, g, x, noise, x_prev, eps_out,
                coef, step_dev, n);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_unipc_step(const float* eps, int cfg, float g, float* x, float* m1, float* m2, float* last, const float* coef,
                               const int32_t* step_dev, int64_t n, pcdm_stream_t s) {
    if (!eps || !x || !m1 || !m2 || !last || !coef || n <= 0) return -1;
    PCDM_LAUNCH(unipc_step_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)s, eps, cfg, g, x, m1, m2, last, coef, step_dev, n);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_unclip_step(const float* pred, int cfg, float g, const float* x, const float* noise, float* x_prev,
                                const float* c8, int64_t n, pcdm_stream_t s) {
    if (!pred || !x || !x_prev || !c8 || n <= 0) return -1;
    UnclipArgs a;
    for (int i = 0; i < 8; ++i) a.c[i] = c8[i];
    PCDM_LAUNCH(unclip_step_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)s, pred, cfg, g, x, noise, x_prev, a, n);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_unclip_step_dev(const float* pred, int cfg, float g, float* x, const float* noise_all, const float* coef,
                                    const int32_t* step_dev, int64_t n, pcdm_stream_t s) {
    if (!pred || !x || !coef || !step_dev || n <= 0) return -1;
    PCDM_LAUNCH(unclip_step_dev_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)s, pred, cfg, g, x, noise_all, coef, step_dev, n);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_lincomb(float* y, int nin, const float* const* xs, const float* c, int64_t n, pcdm_stream_t s) {
    if (!y || nin <= 0 || nin > 6 || !xs || !c || n <= 0) return -1;
    LinArgs a;
    for (int i = 0; i < 6; ++i) {
        a.x[i] = i < nin ? xs[i] : nullptr;
        a.c[i] = i < nin ? c[i] : 0.f;
    }
    PCDM_LAUNCH(lincomb_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)s, y, nin, a, (const float*)nullptr, n);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_rescale_noise_cfg(const float* cfg_eps, const float* text_eps, float* out, int N, int64_t n,
                                      float guidance_rescale, pcdm_stream_t s) {
    if (!cfg_eps || !text_eps || !out || N <= 0 || n <= 1) return -1;
    PCDM_LAUNCH(rescale_cfg_kernel, dim3(N), dim3(1024), 0, (hipStream_t)s, cfg_eps, text_eps, out, n, guidance_rescale);
    PCDM_CHECK_LAUNCH();
    return 0;
}

// bf16 [rows, ldx] -> e4m3 [rows, ldy] (bytes), y = sat(x * scale); 8 elements per thread (16 B in, 8 B out); columns >= cols of a row
// (up to cols_pad) are written as zero: the K / V^T operands of pcdm_flash_attn_fp8
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const u16* __restrict__ x, uint8_t* __restrict__ y, int64_t rows, int cols,
                                                           int cols_pad, int64_t ldx, int64_t ldy, float scale) {
    const int per_row = cols_pad / 8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * per_row) return;
    const int64_t r = i / per_row;
    const int c = (int)(i - r * per_row) * 8;
    float v[8];
    if (c + 8 <= cols && (ldx & 7) == 0) {   // (rows start 16-byte aligned)
        const u16x8 a = *(const u16x8*)(x + r * ldx + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(a[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c + e < cols ? bf2f(x[r * ldx + c + e]) : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e] * scale, -448.f), 448.f);
    u32x2 o = {pack4_fp8(v[0], v[1], v[2], v[3]), pack4_fp8(v[4], v[5], v[6], v[7])};
    *(u32x2*)(y + r * ldy + c) = o;
}

extern "C" int pcdm_quantize_fp8(const void* x, void* y, int64_t rows, int cols, int cols_pad, int64_t ldx, int64_t ldy, float scale,
                                 pcdm_stream_t s) {
    if (!x || !y || rows <= 0 || cols <= 0 || cols_pad < cols || cols_pad % 8 || ldy % 8 || ldy < cols_pad || ldx < cols) return -1;
    const int64_t total = rows * (cols_pad / 8);
    PCDM_LAUNCH(quantize_fp8_kernel, grid1d(total, 256), dim3(256), 0, (hipStream_t)s, (const u16*)x, (uint8_t*)y, rows, cols, cols_pad,
                ldx, ldy, scale);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_softmax_rows(const float* s_in, void* p_out, int rows, int cols, int64_t ld_s, int64_t ld_p, float scale,
                                 pcdm_stream_t s) {
    if (!s_in || !p_out || rows <= 0 || cols <= 0 || cols > 8192) return -1;
    PCDM_LAUNCH(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)s, s_in, (u16*)p_out, cols, ld_s, ld_p,
                scale * 1.44269504088896341f);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_gaussian_sample(const float* moments, const float* noise, float* out, int B, int zc, int HW, float scale,
                                    pcdm_stream_t s) {
    if (!moments || !out || B <= 0 || zc <= 0 || HW <= 0) return -1;
    const int64_t total = (int64_t)B * zc * HW;
    PCDM_LAUNCH(gaussian_sample_kernel, grid1d(total, 256), dim3(256), 0, (hipStream_t)s, moments, noise, out, zc, HW, scale,
                total);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_image_to_uint8(const float* x, void* out, int B, int cstride, int HW, pcdm_stream_t s) {
    if (!x || !out || B <= 0 || cstride < 3 || HW <= 0) return -1;
    const int64_t total = (int64_t)B * HW;
    PCDM_LAUNCH(image_to_uint8_kernel, grid1d(total, 256), dim3(256), 0, (hipStream_t)s, x, (uint8_t*)out, cstride, HW, total);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_pixel_shuffle2(const void* in, void* out, int B, int H, int W, int C, pcdm_stream_t s) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (((uintptr_t)in | (uintptr_t)out) & 15)) return -1;
    const int64_t total = (int64_t)B * 2 * H * 2 * W * (C / 8);
    PCDM_LAUNCH(pixel_shuffle2_kernel, grid1d(total, 256), dim3(256), 0, (hipStream_t)s, (const u16*)in, (u16*)out, H, W, C / 8, total);
    PCDM_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcdm_advance_step(int32_t* step_dev, pcdm_stream_t s) {
    if (!step_dev) return -1;
    PCDM_LAUNCH(advance_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, step_dev);
    PCDM_CHECK_LAUNCH();
    return 0;
}
