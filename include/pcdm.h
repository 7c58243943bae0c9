/* pcdm.h -- C-ABI of libpcdm.so: the MI355X (gfx950) kernels behind the PCDMs stage-2 denoise loop.
 *
 * The reference (tencent-ailab/PCDMs) has NO native/FFI boundary for this path: its boundary is
 * two Python objects, pipe.unet and pipe.scheduler (stage2_batchtest_inpaint_model.py:125-132),
 * and every device op is an ATen/cuDNN/cuBLAS/xformers call made from inside diffusers 0.24.0.
 * Each entry point below therefore cites the reference / diffusers op it replaces (SURVEY.md §2.1
 * K-numbers); the Python mirror of the reference interface lives in pcdms_amd/ and calls these
 * through ctypes.
 *
 * Conventions: plain pointers to DEVICE memory, sizes as ints, a hipStream_t (passed as void*),
 * every call asynchronous on that stream, no allocation inside, no global state.  Activations
 * are NHWC / token-major bf16 ("u16" storage); statistics, biases and time embeddings fp32.
 * Return 0 on success, a negative code otherwise: -1 bad arguments (incl. a tile configuration that is not valid for the
 * problem), -2 an operand beyond the 2 GiB range of the 32-bit buffer offsets, <= -1000 a HIP launch failure (-1000 - hipError_t).
 */
#ifndef PCDM_H
#define PCDM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* pcdm_stream_t; /* hipStream_t */

#define PCDM_ABI_VERSION 5   /* what pcdm_version() returns for the library this header belongs to */
int pcdm_version(void);
/* 1 only for the test-only CPU lane emulator build (tests/emu); the product .so returns 0. */
int pcdm_is_emulator(void);

/* ---- K4 GroupNorm(+SiLU)  [torch.nn.GroupNorm inside ResnetBlock2D.norm1/2, Transformer2DModel.norm,
 *      conv_norm_out: stage2_inpaint_unet_2d_condition.py:435-441,817-819].
 * x = virtual channel-concat of x1 [B,HW,C1] and x2 [B,HW,C2] (x2 may be NULL, C2 = 0): the up-block
 * skip concat (ref :792-793, K6) is never materialised.  y [B,HW,C1+C2] bf16.
 * ws: fp32 workspace of pcdm_groupnorm_ws_floats(B, C1+C2) floats, ZERO-FILLED once when allocated and written by nothing but
 * pcdm_groupnorm afterwards (its head holds arrival counters that every launch leaves at zero); one workspace may serve calls of
 * any shape with B' <= B on one stream.
 * Shapes whose slab is split over several workgroups that exchange statistics inside the launch (UNet levels 0 / 1) rely on those
 * workgroups being co-resident: the library checks the device (occupancy x CUs, no CU mask) and each workgroup's wait is BOUNDED --
 * if its partners do not arrive (~50 ms: CUs held by another stream / process) it computes the slab's statistics alone: slower, same
 * result up to summation order, never a hang.  pcdm_groupnorm_cluster_timeouts reads how often that happened (synchronous). */
int64_t pcdm_groupnorm_ws_floats(int B, int C);
int pcdm_groupnorm_cluster_timeouts(const float* ws, unsigned* count_out, pcdm_stream_t s);
int pcdm_groupnorm(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps,
                   const float* gamma, const float* beta, int fuse_silu, void* y, float* ws, pcdm_stream_t s);

/* GroupNorm whose FIRST source is the not-yet-reduced output of a split-K pcdm_gemm (defer_reduce = 1):
 *   x1[m, n] = bf16( sum_s part[s][m][n] + bias[n] + rowvec[m / HW][n] + residual[m][n] ),  m < M = B * HW, n < N  (= C1)
 * -- the arithmetic (and summation order) of the reduce kernel pcdm_gemm would have launched: results are bit-identical to
 * pcdm_gemm(defer_reduce = 0) + pcdm_groupnorm.  x2 / C2: the second (plain bf16) source of the virtual concat, as pcdm_groupnorm.
 * pre_out [M, N] bf16 must always be provided; it is WRITTEN (the reduced pre-norm tensor, for the residual / skip connections
 * that read it later) iff store_pre != 0 -- except on the two-launch path of very large slabs, which writes it regardless. */
typedef struct pcdm_gn_splitk_src {
    const float* part;     /* pcdm_gemm_params.ws of the producing call */
    int32_t split_k, M, N, Npad;
    const float* bias;     /* [Npad] or NULL */
    const float* rowvec;   /* fp32 [B, ldrv] or NULL (rows_per_batch of the producing call must be HW) */
    int64_t ldrv;
    const int32_t* rowvec_step;   /* as pcdm_gemm_params.rowvec_step / rowvec_step_stride */
    int64_t rowvec_step_stride;
    const void* residual;  /* bf16 [M, ldr] or NULL (res_mod = M) */
    int64_t ldr;
    void* pre_out;
    int32_t store_pre;
    int32_t rowvec_step_count;    /* as pcdm_gemm_params.rowvec_step_count / step_error (ABI 4) */
    int32_t* step_error;
} pcdm_gn_splitk_src;
int pcdm_groupnorm_splitk(const pcdm_gn_splitk_src* src, const void* x2, int C2, int B, int HW, int groups, float eps,
                          const float* gamma, const float* beta, int fuse_silu, void* y, float* ws, pcdm_stream_t s);

/* ---- K8 LayerNorm  [BasicTransformerBlock.norm1/2/3, diffusers attention.py]  x,y [rows,C] bf16 */
int pcdm_layernorm(const void* x, void* y, int rows, int C, float eps, const float* gamma, const float* beta,
                   pcdm_stream_t s);

/* ---- K1/K2/K3/K5/K6/K7/K11 GEMM / implicit-GEMM conv3x3 on MFMA with fused epilogues.
 *  out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )
 *  A (bf16), one of
 *    linear : A[m, k] = (k < c1 ? a[m*lda + k] : a2[m*lda2 + k - c1])      (Linear, 1x1 conv, skip concat)
 *    conv3x3: m = (b, oy, ox) over [B, Ho, Wo]; k = (ky*3+kx)*cin + c; pad 1 (or no_pad_lo); stride 1|2;
 *             upsample=1 reads a[b, iy*Hi/Ho, ix*Wi/Wo, c] (nearest interpolation to Ho x Wo folded in -- iy>>1, ix>>1 for the
 *             usual x2: Upsample2D, K5; other sizes: diffusers' upsample_size path for latents not divisible by 8)
 *  W: packed bf16 [Npad, K] (K contiguous; rows >= N zero).  K % 64 == 0, Npad % 64 == 0.
 *  epilogue: + bias[n] + rowvec[m / rows_per_batch, n] + residual[(m % res_mod), n]  then
 *    PCDM_EPI_STORE   : out[m*ldo + n]                                bf16
 *    PCDM_EPI_GEGLU   : W rows interleaved per 64 as [32 h | 32 gate]; out[m*ldo + n/2..] = h*gelu(gate)
 *    PCDM_EPI_SPLIT_VT: n <  vt_col0 -> out[m*ldo + n];  n >= vt_col0 -> out2[b, n - vt_col0, t]
 *                       with m = b*rows_per_batch + t, out2 pitch ldo2 (V^T for pcdm_flash_attn)
 *    PCDM_EPI_NCHW_F32: out as fp32 [B, N, rows_per_batch] (conv_out -> eps in NCHW) */
enum { PCDM_EPI_STORE = 0, PCDM_EPI_GEGLU = 1, PCDM_EPI_SPLIT_VT = 2, PCDM_EPI_NCHW_F32 = 3 };
enum { PCDM_ACT_NONE = 0, PCDM_ACT_SILU = 1, PCDM_ACT_GELU = 2 };
typedef struct pcdm_gemm_params {
    uint32_t struct_size;  /* = sizeof(pcdm_gemm_params) of the header the HOST was compiled against (ABI 4): a struct of another size -- an older or
                              newer header -- is refused with -1 before any field behind it is read */
    const void* a;
    const void* a2;
    int64_t lda, lda2;
    int32_t c1;
    int32_t conv;      /* 0 linear, 1 conv3x3 */
    int32_t B, Hi, Wi, Ho, Wo, stride, upsample, cin;
    const void* w;
    int32_t M, N, K, Npad;
    const float* bias;
    const float* rowvec;   /* fp32 [M / rows_per_batch, ldrv] */
    int64_t ldrv;          /* 0 -> N */
    int32_t rows_per_batch;
    const void* residual;
    int64_t ldr;
    int32_t res_mod;
    int32_t epilogue;
    int32_t vt_col0;
    void* out;
    int64_t ldo;
    void* out2;
    int64_t ldo2;
    int32_t split_k;   /* > 1: split K over split_k workgroups per tile (PCDM_EPI_STORE only); partial sums in ws */
    float* ws;         /* fp32 workspace, >= split_k * M * Npad floats */
    int64_t ws_floats;
    int64_t ldw;       /* row stride of W in elements (0 -> K); lets an activation slice act as the [N,K] operand */
    int32_t no_pad_lo; /* conv: 1 = zero padding at the bottom/right only (taps start AT the output pixel): the VAE
                          encoder's Downsample2D(padding=0) + F.pad(0,1,0,1); 0 = symmetric padding 1 */
    int32_t tile;      /* 0 = heuristic; 1..26 = explicit tile configuration (gemm.hip dispatch_tile; 22 / 23 = the 176-row tiles), 31..36 = the A-in-registers thin-K
                          kernel (rowgemm.hip; K = 320, linear); -1 if invalid for the problem */
    int32_t act;       /* PCDM_ACT_*: out = act(acc + bias + rowvec) + residual.  With PCDM_EPI_GEGLU: gate activation, 0 = GELU(erf) (GEGLU),
                          PCDM_ACT_SILU = SwiGLU (DINOv2 SwiGLUFFN).  SiLU: the convs of
                          ControlNetConditioningEmbedding (stage2_batchtest_inpaint_model.py:101); GELU(erf): ImageProjModel_p (:54-56) */
    int32_t zero_rows; /* linear only: the caller guarantees A rows [0, zero_rows) are all-zero; they are not read and tiles entirely
                          inside them run the epilogue only.  The CFG unconditional half of attn2.to_out: context == 0 => attention
                          output == 0 => out = bias + residual (stage2_inpaint_pipeline.py:457-458) */
    const float* ln_wsum;  /* non-NULL (tiles 31..36: the A-in-registers kernel, K = 320; tiles 2 / 4 / 7 / 8 / 17 / 18 / 23 / 26: the folded instances of the tiled
                              kernel, any K, see ln_row_stats): W and bias carry a FOLDED LayerNorm -- W' = W diag(gamma),
                              bias' = bias + W beta (BasicTransformerBlock.norm1/2/3 in front of to_q|k|v, to_q and the GEGLU projection: K8 fused into
                              K7 / K11) -- and ln_wsum[n] = sum_k W'[n, k] (fp32 [Npad], of the bf16 values).  The kernel takes each A row's mean / rstd
                              (eps ln_eps) and returns rstd (acc - mean ln_wsum[n]) + bias'[n] = LayerNorm(A) W^T + bias.  Other tiles return -1 when set */
    float ln_eps;
    int32_t defer_reduce;  /* split_k > 1 only: leave the split_k fp32 partial slabs in ws ([split_k][M][Npad]) and do NOT launch the reduce
                              kernel: out is not written; bias / rowvec / residual are not applied.  The consumer applies them:
                              pcdm_groupnorm_splitk (the GroupNorm that follows every split-K convolution of the UNet: conv1 -> norm2, conv2 ->
                              the next block's norm) reads the slabs, so the reduce launch, its bf16 write and the norm's read of it disappear.
                              ws must stay untouched until that consumer has run (same stream). */
    const int32_t* rowvec_step;  /* optional DEVICE counter: the row vector block in use is rowvec + (*rowvec_step) * rowvec_step_stride floats -- the
                                    time-embedding projections of EVERY denoise step are computed once per sampling call (they depend on the
                                    timestep table and the class labels only) and a captured step picks its block by the device step index */
    int64_t rowvec_step_stride;
    int32_t dup_rows;      /* conv3x3 + PCDM_EPI_STORE only, > 0: the M rows computed are ALSO written as rows m + dup_rows of out (which then has
                              M + dup_rows rows), with the row-vector row (m + dup_rows) / rows_per_batch and the residual row m + dup_rows of
                              their own: one contraction, two epilogues.  For a batch whose second half has the same conv INPUT as the first
                              but its own time embedding -- the two classifier-free-guidance halves at conv_in and at the first ResnetBlock2D's
                              conv1 (stage2_inpaint_pipeline.py:499-501 doubles the latents; mask, masked latents and pose are shared).  Needs
                              N % 8 == 0, ldo % 8 == 0, rows_per_batch >= 32 and dup_rows % rows_per_batch == 0 with a rowvec; else -1 */
    const float* ln_row_stats;  /* with ln_wsum on a tiled instance (tiles 2 / 4 / 7 / 8 / 17 / 18 / 26; 23 with partials only): [M][K / 32][2] fp32 -- per A row and 32-column run
                                   {sum, sum of squares about the run's own mean}, as left by the launch that produced A with row_stats_out -- the
                                   kernel merges them into the row's LayerNorm statistics (Chan) instead of taking them in its K loop (NULL: in the
                                   loop, every N tile again: pays for N <~ 1280 only).  K % 64 == 0, K <= 1280, 16-byte aligned */
    float* row_stats_out;       /* linear PCDM_EPI_STORE launches on tiles 2 / 4 / 5 / 6 / 7 / 8 / 10 / 18 (else -1): also write those partials of the
                                   rows stored, [M][N / 32][2] fp32, from the bf16-rounded output values (bias / rowvec / residual included).  The
                                   producer of the rows a LayerNorm reads next (Transformer2DModel.proj_in, attn1 / attn2 .to_out + residual).
                                   N % 32 == 0 */
    int32_t rowvec_step_count;  /* > 0 with rowvec_step: the number of blocks behind rowvec.  A counter value outside [0, count) is CLAMPED into the table on
                                   the device (the launch stays inside the caller's memory) and, if step_error is non-NULL, *step_error is set to 1 -- the
                                   host reads it when it next synchronises (pcdm_unet_step_overflow for the UNet context).  0: unchecked (ABI <= 3 behaviour) */
    int32_t* step_error;        /* DEVICE int32, written only on an out-of-range step (never cleared by the library) */
    const void* a3;             /* conv3x3 with EXTRA K (ABI 5): K = 9 cin + cx, cx > 0 -- behind the nine taps the contraction runs on over a 1x1 convolution of
                                   up to two more NHWC bf16 tensors of the OUTPUT geometry: a2 [M, lda2] supplies the first c1 channels, a3 [M, lda3] the
                                   other cx - c1 (NULL when c1 == cx); W rows are [9 cin taps | c1 | cx - c1].  ResnetBlock2D.conv_shortcut over the block's
                                   (concatenated) input composed into conv2 (resnet.py as composed at stage2_inpaint_unet_2d_condition.py:321-344,407-430):
                                   no shortcut launch, no residual round trip.  stride 1, no upsample, Hi == Ho, Wi == Wo, symmetric padding, no dup_rows;
                                   c1, cx multiples of 64; lda2 >= c1, lda3 >= cx - c1, both multiples of 8; else -1 */
    int64_t lda3;
    uint64_t tap_lut;           /* conv3x3 over a SUBSET of the nine taps per output-channel group (ABI 5): with tap_group_n > 0 the K axis holds ntaps = K / cin
                                   (1..4) taps; output channels [g tap_group_n, (g + 1) tap_group_n) (g < 4) use the taps whose ids (ky * 3 + kx, 0..8) are the
                                   nibbles of bits [16 g, 16 g + 4 ntaps) of tap_lut, in K order; W rows are [ntaps taps x cin].  The phase decomposition of
                                   Upsample2D: conv3x3(nearest-upsample x2 (x)) at output pixel (2y + a, 2x + b) is a 2x2 convolution of x with summed taps, i.e.
                                   one 3x3 launch on the LOW-RES input with N = 4 Cout (group = phase 2a + b, taps {3(a+i) + b + j}), 4/9 of the FLOPs, followed by
                                   pcdm_pixel_shuffle2 (diffusers Upsample2D as composed at stage2_inpaint_unet_2d_condition.py:407-430).  stride 1, no upsample,
                                   no extra K (a2 / a3), tap_group_n a multiple of the N tile in use (else -1: pick a tile that divides it) */
    int32_t tap_group_n;
} pcdm_gemm_params;
/* pcdm_version() == 5: the struct above STARTS with struct_size and ends with a3, lda3, tap_lut, tap_group_n (4: ended with rowvec_step_count, step_error; 3: no struct_size, ended with
 * ln_row_stats, row_stats_out, gn_stats_out, gn_stats_gs -- the last two are gone with the GroupNorm-statistics producer; 2: ended with
 * dup_rows; 1: with ln_eps).  Zero-initialise it (memset), set struct_size = sizeof(pcdm_gemm_params): the library compares it with its own and
 * returns -1 on a mismatch, so a host built against another header fails at its first call instead of having trailing fields misread.
 * (Comparing pcdm_version() with PCDM_ABI_VERSION at start-up remains good practice: the other entry points have no such guard.)
 * bias, rowvec, ldrv and rowvec_step_stride must keep 16-byte alignment (4 floats): the epilogues load them as float4; a violation returns -1.
 * *rowvec_step is device memory the library cannot validate at launch: pass rowvec_step_count to have it bounded on the device. */
int pcdm_gemm(const pcdm_gemm_params* p, pcdm_stream_t s);

/* ---- K9/K10 fused attention (replaces xformers.ops.memory_efficient_attention enabled at
 *      stage2_batchtest_inpaint_model.py:133).  head_dim = 64.  softmax(q k^T * scale) v, fp32 softmax.
 *  q  [B*Lq, ldq]  bf16, head h at columns [64h, 64h+64)
 *  k  [B*Lk, ldk]  bf16, same head layout
 *  vt [B, H*64, ldvt] bf16 = V transposed (key index contiguous), as written by PCDM_EPI_SPLIT_VT
 *  o  [B*Lq, ldo]  bf16 */
int pcdm_flash_attn(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                    void* o, int64_t ldo, int B, int H, int Lq, int Lk, float scale, pcdm_stream_t s);
/* The same with the lazy-rescale threshold made explicit: the running softmax reference of a query moves only when a key tile's
 * maximum exceeds it by more than thr_log2 (log2 units, 0 <= thr <= 16; P <= 2^thr); 0 = eager online softmax.  pcdm_flash_attn
 * uses PCDM_ATTN_DEFAULT_THR.  Mathematically identical for every thr (P, the row sum and O carry the same reference). */
#define PCDM_ATTN_DEFAULT_THR 8.0f
int pcdm_flash_attn_thr(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                        void* o, int64_t ldo, int B, int H, int Lq, int Lk, float scale, float thr_log2, pcdm_stream_t s);

/* ---- N4 (BASELINE.json configs[4]): the same attention with OCP e4m3 operands on the MX-scaled fp8 MFMA (unit block
 *      scales; twice the bf16 matrix rate).  No reference counterpart (attention enters at stage2_batchtest_inpaint_model.py:133).
 * pcdm_quantize_fp8: y[r, c] = e4m3(sat(x[r, c] * scale)) for bf16 x [rows, ldx] -> bytes y [rows, ldy]; columns [cols, cols_pad) are
 *   written as zero (cols_pad % 8 == 0).  Used for K [B*Lk, C] and V^T [B*C, Lk -> padded to a multiple of 16].
 * pcdm_flash_attn_fp8: q bf16 as in pcdm_flash_attn; k8 [B*Lk, ldk] / vt8 [B, H*64, ldvt] e4m3 bytes (ldk, ldvt multiples of 16,
 *   ldvt >= Lk, padding zero); k_descale / v_descale undo the quantisation scales; thr_log2 <= 8.  fp32 softmax, P rounded to e4m3. */
int pcdm_quantize_fp8(const void* x, void* y, int64_t rows, int cols, int cols_pad, int64_t ldx, int64_t ldy, float scale, pcdm_stream_t s);
int pcdm_flash_attn_fp8(const void* q, int64_t ldq, const void* k8, int64_t ldk, const void* vt8, int64_t ldvt, void* o, int64_t ldo,
                        int B, int H, int Lq, int Lk, float scale, float k_descale, float v_descale, float thr_log2, pcdm_stream_t s);

/* ---- K12 time / class embedding helpers.
 * pcdm_timestep_embedding: diffusers Timesteps(dim, flip_sin_to_cos, shift) (ref :184,677): out fp32 [B,dim];
 *   t read from DEVICE memory: t_dev[step_dev ? *step_dev : 0] (int64), broadcast over B.
 * pcdm_small_linear: y[b,n] = act_out( sum_k act_in(x[b,k]) * W[n,k] + bias[n] ), x,y fp32, W bf16 [N,K],
 *   B <= 32; act_in: 1 = SiLU; act_out: 1 = SiLU before `add`, 2 = SiLU after `add`.  (TimestepEmbedding MLPs ref :191-197,247; every ResnetBlock2D.time_emb_proj) */
int pcdm_timestep_embedding(const int64_t* t_dev, const int32_t* step_dev, float* out, int B, int dim,
                            int flip_sin_to_cos, float shift, pcdm_stream_t s);
int pcdm_small_linear(const float* x, const void* w, const float* bias, const float* add, float* y, int B, int K,
                      int N, int act_in, int act_out, pcdm_stream_t s);
/* The same embeddings for a whole timestep TABLE (once per sampling call instead of five launches per denoise step):
 * pcdm_timestep_embedding_rows: out[i, :] = Timesteps(t_dev[i]), i < n (fp32 [n, dim]);
 * pcdm_time_class_combine: out[i * B + b, :] = bf16( silu( emb_t[i, :] + (cls ? cls[b, :] : 0) ) ) -- emb = time_embedding(t_i) + class_embedding(b)
 *   as every ResnetBlock2D consumes it (silu(emb)), for all steps and batch entries: the A operand of ONE time_emb_proj GEMM with
 *   M = n * B rows.  Same per-row arithmetic as the per-step launches (bit-identical results). */
int pcdm_timestep_embedding_rows(const int64_t* t_dev, int n, float* out, int dim, int flip_sin_to_cos, float shift, pcdm_stream_t s);
int pcdm_time_class_combine(const float* emb_t, const float* cls, void* out_bf16, int n, int B, int D, pcdm_stream_t s);

/* ---- P-2 input assembly: cat([cat([latents]*2), mask, masked_latents], 1) (stage2_inpaint_pipeline.py:499-501)
 * -> NHWC bf16 [Bout, h, w, cpad] with channels >= 9 zero.  latents fp32 NCHW [N,4,h,w]; Bout = rep*N rows
 * (rep = 2 with CFG); mask fp32 [1|Bout,1,h,w]; masked fp32 [1|Bout,4,h,w] (batch-broadcast when *_b == 1).
 * mask == NULL: the 8-channel stage-3 input cat([latents, gen_t_img_latents], 1) (stage3_refined_pipeline.py:538). */
int pcdm_assemble_input(const float* latents, int N, int rep, const float* mask, int mask_b, const float* masked,
                        int masked_b, void* out, int h, int w, int cpad, pcdm_stream_t s);
/* NCHW fp32 -> NHWC bf16 (pose feature st_pose_f, ref :430-431) and back. */
int pcdm_nchw_f32_to_nhwc_bf16(const float* x, void* y, int B, int C, int Cpad, int HW, pcdm_stream_t s); /* y [B,HW,Cpad], c >= C zero */
int pcdm_nhwc_bf16_to_nchw_f32(const void* x, float* y, int B, int C, int HW, pcdm_stream_t s);
int pcdm_f32_to_bf16(const float* x, void* y, int64_t n, pcdm_stream_t s);

/* ---- K13 CFG combine + scheduler step (stage2_inpaint_pipeline.py:510-519).
 * eps [2N or N, C*HW] fp32 (uncond rows first).  g = guidance scale (cfg=0: eps used as is).
 * x_prev = cx*x + ce*eps_guided (+ cn*noise); coefficients read from DEVICE table coef[step][4] =
 * {cx, ce, cn, unused} at index *step_dev so that one captured hipGraph serves every step.
 * Optionally writes the guided eps (eps_out) and x0 = c0x*x + c0e*eps is left to the host scheduler. */
int pcdm_cfg_step(const float* eps, int cfg, float g, const float* x, const float* noise, float* x_prev,
                  float* eps_out, const float* coef, const int32_t* step_dev, int64_t n, pcdm_stream_t s);
/* CFG combine + diffusers UniPCMultistepScheduler.step (the shipped driver's scheduler: stage2_batchtest_inpaint_model.py:132;
 * SURVEY.md Appendix A-10; solver order <= 2) as one kernel on static state, replayable from a hipGraph.  Row *step_dev of the DEVICE
 * table coef[step][12] = {a_x, a_e, use_corrector, c_last, c_m1, c_m2, c_mt, p_x, p_mt, p_m1, 0, 0}:
 *   m_t = a_x x + a_e eps_guided;  x_c = use_corrector ? c_last last + c_m1 m1 + c_m2 m2 + c_mt m_t : x;  x' = p_x x_c + p_mt m_t + p_m1 m1
 * then, in place: x <- x', m2 <- m1, m1 <- m_t, last <- x_c (all fp32 [n]; zero m1 / m2 / last before step 0). */
int pcdm_unipc_step(const float* eps, int cfg, float g, float* x, float* m1, float* m2, float* last, const float* coef,
                    const int32_t* step_dev, int64_t n, pcdm_stream_t s);
/* Stage-1 prior: CFG combine (src/pipelines/stage1_prior_pipeline.py:467-471) + diffusers
 * UnCLIPScheduler.step (:478-483) + optional affine read-out (post_process_latents, stage1_prior_transformer.py:299-301).
 * pred [2N or N, n/N] fp32 (uncond rows first); HOST coefficients c8 = {p_x, p_e, clip, c_x0, c_x, c_noise, out_scale,
 * out_shift}:  x0 = clamp(p_x*x + p_e*pred_guided, +-clip) (clip <= 0: none);
 * x_prev = (c_x0*x0 + c_x*x + c_noise*noise) * out_scale + out_shift.  x_prev may alias x. */
int pcdm_unclip_step(const float* pred, int cfg, float g, const float* x, const float* noise, float* x_prev,
                     const float* c8, int64_t n, pcdm_stream_t s);
/* The same step with its eight coefficients read from row *step_dev of a DEVICE table coef[steps][8] and its noise from slab *step_dev
 * of noise_all [steps, n] (NULL: none), updating x in place: one captured hipGraph serves every step of the stage-1 loop. */
int pcdm_unclip_step_dev(const float* pred, int cfg, float g, float* x, const float* noise_all, const float* coef,
                         const int32_t* step_dev, int64_t n, pcdm_stream_t s);
/* rescale_noise_cfg (stage2_inpaint_pipeline.py:52-63): out = gr * cfg * std(text)/std(cfg) + (1-gr) * cfg, per sample
 * over n = C*H*W elements (unbiased std); cfg_eps / text_eps / out fp32 [N, n]; out may alias cfg_eps. */
int pcdm_rescale_noise_cfg(const float* cfg_eps, const float* text_eps, float* out, int N, int64_t n,
                           float guidance_rescale, pcdm_stream_t s);
/* p[r, c] = softmax_c(scale * s[r, c]) : fp32 [rows, ld_s] -> bf16 [rows, ld_p], cols <= 8192.  The VAE's single-head
 * d = 512 attention (AutoencoderKL mid block) = pcdm_gemm (K Q^T, fp32) + this + pcdm_gemm (P V). */
int pcdm_softmax_rows(const float* s_in, void* p_out, int rows, int cols, int64_t ld_s, int64_t ld_p, float scale,
                      pcdm_stream_t s);
/* AutoencoderKL helpers (stage2_inpaint_pipeline.py:443-444, :528-532):
 * gaussian_sample: out[B,zc,HW] = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale from moments fp32 [B,2*zc,HW];
 * image_to_uint8: VaeImageProcessor.postprocess -- x fp32 [B,cstride,HW] (first 3 channels) -> uint8 [B,HW,3]. */
int pcdm_gaussian_sample(const float* moments, const float* noise, float* out, int B, int zc, int HW, float scale,
                         pcdm_stream_t s);
int pcdm_image_to_uint8(const float* x, void* out, int B, int cstride, int HW, pcdm_stream_t s);
/* y = sum_i c[i] * x_i  (i < nin <= 6), fp32; UniPC predictor/corrector linear combinations. */
int pcdm_lincomb(float* y, int nin, const float* const* xs, const float* c, int64_t n, pcdm_stream_t s);
/* *step_dev += 1 */
int pcdm_advance_step(int32_t* step_dev, pcdm_stream_t s);
/* out NHWC bf16 [B, 2H, 2W, C] <- in [B * H * W, 4 * C] bf16 (columns = phase 2a + b major, then channel): out[b, 2y + a, 2x + bb, c] =
 * in[(b, y, x), (2a + bb) C + c].  The second half of the phase-decomposed Upsample2D convolution (pcdm_gemm_params.tap_lut).  C % 8 == 0. */
int pcdm_pixel_shuffle2(const void* in, void* out, int B, int H, int W, int C, pcdm_stream_t s);

/* ---- The UNet forward as ONE entry ("a fused unet_forward(ctx, ...)" over an opaque context).
 * Replaces Stage2_InapintUNet2DConditionModel.forward (/root/reference/src/models/stage2_inpaint_unet_2d_condition.py:579-825) for a host
 * that is not Python: pcdm_unet_create from the topology, pcdm_unet_set_weight / _set_vector with the packed tensors under their
 * diffusers module paths, ONE caller-owned workspace (pcdm_unet_workspace_bytes, zeroed once by pcdm_unet_workspace_init), then per
 * sampling call pcdm_unet_prepare_conditioning (class embedding, pose feature, cross-attention K / V^T: step-invariant) and per denoise
 * step pcdm_unet_forward.  No allocation inside, every launch on the caller's stream, fixed scratch addresses (hipGraph-capturable).
 * Weight names (N = diffusers module path, e.g. "down_blocks.0.resnets.1." / "...attentions.0."):
 *   optional "up_blocks.i.upsamplers.0.conv4" (the upsampler's convolution as four 2x2 phase kernels: pcdm_gemm_params.tap_lut; used when the target is 2H x 2W),
 *   conv_in, conv_out, N"conv1", N"conv2", N"conv_shortcut", optional N"conv2s" (conv2's packed rows with conv_shortcut's [N, Cx] appended along K,
 *   bias = the sum: when registered, conv2 + conv_shortcut of that resnet run as ONE launch through pcdm_gemm_params.a2 / a3), "down_blocks.i.downsamplers.0.conv", "up_blocks.i.upsamplers.0.conv"  (pcdm_pack conv3x3 / linear)
 *   N"proj_in", N"proj_out", N"qkv" (to_q|to_k|to_v rows), N"o1", N"q2", N"kv2" (to_k|to_v of attn2), N"o2", N"ff1" (GEGLU packing), N"ff2",
 *   optional N"qkv_ln" / N"q2_ln" / N"ff1_ln" (LayerNorm folded, with wsum), optional N"ffo" = [Wp W2 | Wp] with bias Wp b2 + bp (W2, b2 = ff.net.2;
 *   Wp, bp = proj_out; N = C, K = 5 C: when registered, ff.net.2 (+ residual) -> proj_out (+ residual) of that block run as ONE two-source GEMM),
 *   "time_emb_proj" (all resnets' rows concatenated in resnet order),
 *   "time_embedding.linear_1/2", "class_embedding.linear_1/2" (bf16 [N, K] unpadded + fp32 bias, for pcdm_small_linear)
 * Vectors (fp32): N"norm1.weight" / ".bias", N"norm2.*" (resnets), N"norm.*" and N"transformer_blocks.0.norm{1,2,3}.*" (transformers),
 *   "conv_norm_out.weight" / ".bias".
 * Tile choices: a new context starts with the committed tuning table (pcdms_amd/tuning/gfx950.json, measured on MI355X; compiled in through
 * csrc/tuning_table.inc) -- the (tile, split_k) pcdm_gemm should use for a problem key (ln, M, Npad, K, conv, stride, upsample, epilogue,
 * two_source, residual, flag: 1 = zero_rows, 2 = dup_rows); pcdm_unet_set_tile overrides / adds entries, pcdm_unet_get_tile reads one (-1: none:
 * the library heuristic decides).
 * Return codes as everywhere; pcdm_unet_last_error names the missing weight / failing call. */
typedef struct pcdm_unet pcdm_unet;
typedef struct pcdm_unet_config {
    int32_t out_channels;
    int32_t n_levels;               /* <= 8 */
    int32_t block_out_channels[8];  /* multiples of 64; heads[i] * 64 == block_out_channels[i] */
    int32_t heads[8];
    int32_t cross_attn[8];          /* down block i is CrossAttnDownBlock2D (up block n-1-i mirrors it) */
    int32_t layers_per_block;
    int32_t cross_attention_dim;
    int32_t norm_groups;
    float norm_eps;
    int32_t class_embed;            /* 1: "projection" class embedding (stage 2), 0: none (stage 3) */
    int32_t flip_sin_to_cos;
    float freq_shift;
} pcdm_unet_config;
pcdm_unet* pcdm_unet_create(const pcdm_unet_config* cfg);
void pcdm_unet_destroy(pcdm_unet* u);
const char* pcdm_unet_last_error(const pcdm_unet* u);
int pcdm_unet_set_weight(pcdm_unet* u, const char* name, const void* w_bf16, const float* bias, const float* wsum, int N, int K, int Npad, int cin);
int pcdm_unet_set_vector(pcdm_unet* u, const char* name, const float* v, int n);
int pcdm_unet_set_tile(pcdm_unet* u, int ln, int M, int Npad, int K, int conv, int stride, int upsample, int epilogue, int two_source, int residual,
                       int zero_rows, int tile, int split_k);
/* 1: every attention of the UNet with OCP e4m3 K / V^T / Q / P operands (pcdm_quantize_fp8 + pcdm_flash_attn_fp8; BASELINE.json configs[4]), 0
 * (default): bf16.  Switch before pcdm_unet_prepare_conditioning (it quantises the context K / V^T) and before capturing a graph. */
int pcdm_unet_set_attention_fp8(pcdm_unet* u, int on);
int pcdm_unet_get_tile(const pcdm_unet* u, int ln, int M, int Npad, int K, int conv, int stride, int upsample, int epilogue, int two_source, int residual,
                       int flag, int* tile, int* split_k);
int64_t pcdm_unet_workspace_bytes(pcdm_unet* u, int B, int h, int w, int L);
int pcdm_unet_workspace_init(pcdm_unet* u, int B, int h, int w, int L, void* workspace, pcdm_stream_t s);
/* ehs fp32 [B, L, cross_attention_dim]; class_labels fp32 [B, K of class_embedding.linear_1] or NULL; pose fp32 NCHW [pose_b = 1 | B, C0, h, w] or
 * NULL; zero_ctx_batches: leading batch entries of ehs that are all-zero (the CFG unconditional half: their cross-attention is skipped) */
int pcdm_unet_prepare_conditioning(pcdm_unet* u, int B, int h, int w, int L, const float* ehs, const float* class_labels, const float* pose, int pose_b,
                                   int zero_ctx_batches, void* workspace, pcdm_stream_t s);
/* Optional, after pcdm_unet_prepare_conditioning on the same workspace: the caller guarantees that batch entries b and b + B/2 always carry the
 * same x_in rows and pose feature (the two classifier-free-guidance halves: stage2_inpaint_pipeline.py:499-501 doubles the latents and shares
 * mask / masked latents / pose): conv_in, the first norm1 and the first conv1's contraction then run once for both halves. */
int pcdm_unet_set_shared_cfg_input(pcdm_unet* u, void* workspace, int shared);
/* Optional, after pcdm_unet_prepare_conditioning on the same workspace: the time / class embedding MLPs (ref :661-708) and every
 * ResnetBlock2D.time_emb_proj for ALL n timesteps of t_dev, once per sampling call (2 + 2 ceil(n / 32) + 2 launches) instead of five launches
 * per denoise step.  table: caller-owned device memory of pcdm_unet_time_table_bytes(u, n, B) bytes, alive while forwards use it.
 * pcdm_unet_forward calls on this workspace that pass the same t_dev and a device step counter pick their block by that counter; any other
 * call computes the embeddings per step as before.  Bit-identical to the per-step launches.  The table is keyed on the POINTER t_dev: a host
 * that rewrites the timestep buffer in place (another schedule or step count) must call pcdm_unet_prepare_timesteps again before the
 * next forward (not checkable at launch: it lives in device memory).  The device step counter must stay below n: ABI 4 bounds it on the
 * device (clamped into the table, error flag: pcdm_unet_step_overflow). */
int64_t pcdm_unet_time_table_bytes(const pcdm_unet* u, int n, int B);
int pcdm_unet_prepare_timesteps(pcdm_unet* u, const int64_t* t_dev, int n, void* table, void* workspace, pcdm_stream_t s);
/* x_in NHWC bf16 [B, h, w, conv_in.cin] (pcdm_assemble_input / pcdm_nchw_f32_to_nhwc_bf16); timestep = t_dev[step_dev ? *step_dev : 0] (device);
 * pose_b as passed to prepare_conditioning (0: no pose); eps_out fp32 NCHW [B, out_channels, h, w] */
int pcdm_unet_forward(pcdm_unet* u, const void* x_in, const int64_t* t_dev, const int32_t* step_dev, int B, int h, int w, int L, int pose_b,
                      void* workspace, float* eps_out, pcdm_stream_t s);
/* Did a pcdm_unet_forward on this workspace ever find *step_dev outside the time table of pcdm_unet_prepare_timesteps (>= n or negative)?  Such a
 * forward does NOT read out of bounds -- every consumer of the table clamps the step on the device -- but its time embedding is that of the
 * last table row: *flag_out = 1 then (and stays 1 until pcdm_unet_workspace_init / pcdm_unet_prepare_timesteps clear it).  Synchronises s. */
int pcdm_unet_step_overflow(pcdm_unet* u, void* workspace, int* flag_out, pcdm_stream_t s);
/* Host-side weight packing (plain loops on HOST memory; upload the result): the layouts pcdm_gemm / pcdm_unet_set_weight expect, from the
 * fp32 tensors of a diffusers state dict.  Each returns Npad (< 0: bad arguments); output pointers may be NULL to query sizes.
 *   pcdm_pack_linear : [N, K] -> bf16 [Npad, K], bias -> fp32 [Npad]                       (Npad = N rounded up to pad_to, normally 64)
 *   pcdm_pack_conv3x3: [N, Cin, 3, 3] -> bf16 [Npad, 9 Cp], k = (ky 3 + kx) Cp + c          (Cp = Cin rounded up to 64; *K_out = 9 Cp, *cin_out = Cp)
 *   pcdm_pack_geglu  : [2 D, K] rows [h | gate] -> bf16 [2 Dp, K], rows per 64 = [32 h | 32 gate], bias likewise (GEMM N = D, Npad = 2 Dp) */
int pcdm_pack_linear(const float* w, const float* bias, int N, int K, int pad_to, uint16_t* out_w, float* out_bias);
int pcdm_pack_conv3x3(const float* w, const float* bias, int N, int Cin, int pad_to, uint16_t* out_w, float* out_bias, int* K_out, int* cin_out);
int pcdm_pack_geglu(const float* w, const float* bias, int D, int K, uint16_t* out_w, float* out_bias);

#ifdef __cplusplus
}
#endif
#endif
