// Extended instances of the tiled bf16 MFMA GEMM (gemm_kernel.inc, template parameter EXT) for the LayerNorm -> Linear pairs of the UNet's
// levels 1-3 (K = 640 / 1280: BasicTransformerBlock.norm1 -> to_q|k|v, norm2 -> attn2.to_q, norm3 -> the GEGLU projection;
// /root/reference/src/models/stage2_inpaint_unet_2d_condition.py:321-361,407-430 -> diffusers BasicTransformerBlock), round 5:
//
//   EXT = 3  producer: the linear that WRITES the rows a LayerNorm reads next (proj_in, attn1.to_out + residual, attn2.to_out + residual) also
//            leaves, per row, the {sum, M2 about the run's own mean} of every 32-column run of the bf16 values it stores
//            (pcdm_gemm_params.row_stats_out [M][N / 32][2]; 4 lanes per run, two xor-shuffles per sum, one 8-byte store);
//   EXT = 2  consumer: the weights carry the FOLDED LayerNorm (W' = W diag(gamma), b' = b + W beta, ln_wsum = rowsum(W')); the prologue merges
//            the row's K / 32 partials with Chan's formula into {mean, rstd} (LDS), and the accumulators become rstd (acc - mean wsum[n]) where
//            they are staged for the epilogue: out = LayerNorm(A) W^T + b with NO LayerNorm launch, no normalised tensor, and no
//            statistics work in the K loop;
//   EXT = 1  consumer without a producer (any caller that only has the rows): the statistics are taken in the K loop from the A tiles as
//            they pass through LDS (shifted sums) -- every N tile of a row block redoes them, so this form pays only for N <~ 1280
//            (profiles/r5_bench_ln_gemm.txt: to_q 18.3 -> 13.7 us, but to_q|k|v 51.3 -> 57.8 us).
// A separate translation unit so that these ~20 instantiations compile beside gemm.hip's, not behind them.
#include "gemm_kernel.inc"

namespace {
template <int EXT>
int dispatch_ext(int tile, const GemmArgs& a, hipStream_t st) {
    switch (tile) {
        case 2: return launch_gemm<64, 64, 2, 2, 2, false, false, 32, 64, EXT>(a, st);
        case 4: return launch_gemm<128, 128, 2, 2, 2, false, false, 32, 64, EXT>(a, st);
        case 7: return launch_gemm<128, 128, 2, 2, 3, false, false, 32, 64, EXT>(a, st);
        case 8: return launch_gemm<64, 64, 2, 2, 4, false, false, 32, 64, EXT>(a, st);
        case 18: return launch_gemm<128, 128, 4, 2, 2, false, false, 32, 64, EXT>(a, st);
        default: break;
    }
    if constexpr (EXT == 3) {        // producers: the N-narrow tiles the residual linears of level 2 are tuned to
        switch (tile) {
            case 5: return launch_gemm<128, 64, 2, 2, 2, false, false, 32, 64, EXT>(a, st);
            case 6: return launch_gemm<256, 64, 4, 2, 3, false, false, 32, 64, EXT>(a, st);
            case 10: return launch_gemm<128, 64, 2, 2, 3, false, false, 32, 64, EXT>(a, st);
            default: return -1;
        }
    } else {                         // consumers: the wide tiles of the N >= 1920 projections
        switch (tile) {
            case 17: return launch_gemm<256, 256, 2, 4, 2, false, false, 32, 64, EXT>(a, st);
            case 26: return launch_gemm<192, 256, 2, 4, 2, false, false, 16, 64, EXT>(a, st);
            case 23:   // round 6: the 176-row tile (M = 11264 = 64 x 176: level 1's GEGLU projection fills the chip exactly); partials only
                if constexpr (EXT == 2) return launch_gemm<176, 256, 2, 4, 2, false, false, 16, 64, EXT>(a, st);
                else return -1;
            default: return -1;
        }
    }
}
}  // namespace

int pcdm_gemm_detail::launch_gemm_ext(int ext, int tile, const GemmArgs& a, hipStream_t st) {
    switch (ext) {
        case 1: return dispatch_ext<1>(tile, a, st);
        case 2: return dispatch_ext<2>(tile, a, st);
        case 3: return dispatch_ext<3>(tile, a, st);
        default: return -1;
    }
}
