// Kernel-argument block shared by the GEMM kernels (gemm.hip: tiled GEMM / implicit-GEMM conv; rowgemm.hip: the A-in-registers
// thin-K kernel) and the few epilogue helpers both use.
#pragma once
#include "pcdm_device.h"
#include "../../include/pcdm.h"

namespace pcdm_gemm_detail {
struct GemmArgs {
    const u16* a;
    const u16* a2;
    int64_t lda, lda2;
    int c1;
    const u16* a3;     // conv with extra K (K = 9 cin + cx): the 1x1 sources behind the taps are a2 (c1 channels) and a3 (cx - c1); pcdm_gemm_params.a3
    int64_t lda3;
    uint64_t tap_lut;  // conv over a SUBSET of the nine taps per output-channel group (pcdm_gemm_params.tap_lut / tap_group_n): 4 groups x 4 nibbles of true tap ids
    int tap_group_n;   // output channels per group (a multiple of the N tile in use); 0: the plain nine taps
    uint32_t xinv;     // ceil(2^31 / (cin / 64)): pixel index of a conv row from its centre-tap offset (gemm_kernel.inc issue_tile)
    int B, Hi, Wi, Ho, Wo, stride, upsample, cin;
    int pad;   // conv: 1 = symmetric zero padding 1 (default); 0 = bottom/right only (VAE encoder Downsample2D)
    const u16* w;
    int64_t ldw;   // row stride of W in elements (>= K)
    int M, N, K, Npad;
    const float* bias;
    const float* rowvec;
    const int32_t* rowvec_step;     // device counter selecting the row-vector block: rowvec + *rowvec_step * rowvec_step_stride (pcdm_gemm_params)
    int64_t rowvec_step_stride;
    int rowvec_step_count;          // > 0: blocks behind rowvec; a counter outside [0, count) is clamped and *step_error set (pcdm_gemm_params, ABI 4)
    int32_t* step_error;
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
    int split_k;   // > 1: K range split over split_k workgroups per tile, fp32 partials to ws, reduced by splitk_reduce_kernel
    float* ws;
    int act;    // PCDM_ACT_*: applied to (acc + bias + rowvec), before the residual add
    int zero_rows;  // linear only: A rows < zero_rows are all-zero and are never read (tiles entirely inside skip their main loop)
    const float* ln_wsum;    // rowgemm tiles and gemm_ext.hip (EXT 1 / 2): the weights carry a folded LayerNorm (W diag(gamma), bias + W beta); fp32 [Npad] row sums of
    float ln_eps;            // the folded weights: out = rstd (acc - mean wsum[n]) + bias[n] with the row's own mean / rstd (eps ln_eps)
    const float* ln_row_stats;   // folded LayerNorm, tiled instances (gemm_ext.hip): [M][K / 32][2] partial {sum, M2} of the A rows, written by their producer
    float* row_stats_out;        // row-statistics producer instances: [M][N / 32][2] partials of the stored rows
    int dup_rows;       // conv, lean epilogue: also write rows m + dup_rows (their own rowvec / residual rows): pcdm_gemm_params.dup_rows
    int defer_reduce;   // split_k > 1: no reduce launch (pcdm_groupnorm_splitk consumes the partial slabs)
    int debug;  // ablation (tools/ablate_gemm.py): bit0 = skip steady-state loads, bit1 = skip MFMAs (staggered tiles only), bit2 = per-workgroup
                // phase time stamps (s_memtime) into ws[wg][8] as uint64 (tools/gemm_anatomy.py)
};
}  // namespace pcdm_gemm_detail

namespace pcdm_gemm_detail {
// the row-vector block of this launch (one scalar load when a step counter is given)
// the device step counter bounded into a table of `count` blocks (count <= 0: unchecked); an out-of-range value is reported through *err
// (every lane that sees it stores the same 1: a benign race) and the launch reads the nearest valid block instead of foreign memory
__device__ __forceinline__ int bounded_step(const int32_t* step, int count, int32_t* err) {
    int st = *step;
    if (count > 0 && (st < 0 || st >= count)) {
        if (err) *err = 1;
        st = st < 0 ? 0 : count - 1;
    }
    return st;
}
__device__ __forceinline__ const float* rowvec_base(const GemmArgs& p) {
    return (p.rowvec && p.rowvec_step) ? p.rowvec + (int64_t)bounded_step(p.rowvec_step, p.rowvec_step_count, p.step_error) * p.rowvec_step_stride : p.rowvec;
}
__device__ __forceinline__ float apply_act(float v, int act) {
    return act == PCDM_ACT_SILU ? silu_f(v) : act == PCDM_ACT_GELU ? gelu_erf_f(v) : v;
}
// gated-linear-unit epilogue: GEGLU (diffusers FeedForward, act == 0) or SwiGLU (DINOv2 SwiGLUFFN, act == PCDM_ACT_SILU)
__device__ __forceinline__ float gate_act(float g, int act) { return act == PCDM_ACT_SILU ? silu_f(g) : gelu_erf_f(g); }

// tile ids >= kRowGemmTile0 of pcdm_gemm_params.tile: rowgemm.hip (returns -1 when the problem / epilogue is not one it takes)
constexpr int kRowGemmTile0 = 31;
int launch_rowgemm(int tile, const GemmArgs& a, hipStream_t st);
// gemm_ext.hip: the extended instances of the tiled kernel (gemm_kernel.inc, template EXT): 1 / 2 = folded-LayerNorm consumers (row statistics in
// the K loop / from the producer's partials), 3 = row-statistics producers.  Returns -1 for a tile id without such an instance.
int launch_gemm_ext(int ext, int tile, const GemmArgs& a, hipStream_t st);
}  // namespace pcdm_gemm_detail
