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
#include "gemm_kernel.inc"

namespace {
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
        // Round 6 -- 176-row tiles: 45056 = 256 x 176, 11264 = 64 x 176, 2816 = 16 x 176, 704 = 4 x 176: every level of the 64 x 88 latent
        // (5632 = 32 x 176 rows per image) divides into 176-row tiles, so M tiles x N tiles x split-K lands on the 256 CUs EXACTLY where the
        // 192-row tiles give 235 / 236 / 240 workgroups of 9 % more rows each.  176 = 11 fragment rows of 16: the first wave row takes 6 (96
        // pixels), the second 5 (80) -- its own instance of the K loop (gemm_kernel.inc UNEVEN); waves w and w + 4 share a SIMD, so every SIMD
        // carries 6 + 5.  Same LDS (the A stage keeps 192 row slots, 16 of them never fetched), same registers, same schedule as 21 / 26.
        case 22: return launch_gemm<176, 320, 2, 4, 2, CONV, false, 16>(a, st);   // 8 waves, 128 KiB, 1 block / CU
        case 23: return launch_gemm<176, 256, 2, 4, 2, CONV, false, 16>(a, st);   // 8 waves (96|80 x 64: GEGLU-capable), 112 KiB
        // (Round 6 also measured 16x16x32-fragment twins of tiles 17 / 1 / 18 / 4 -- ids 24 / 25 / 29 / 30; the instruction is ~9 % more energy-efficient on
        //  this power-limited part: 4-15 % FASTER per launch back to back on the N = 640 / 1280 / 5120 shapes and 0.15-0.5 % SLOWER end to end in three
        //  interleaved same-box A/Bs (profiles/r6_tile_f16.txt, r6_ab_tiles.json): removed.  Back-to-back launches of one problem -- weights resident
        //  in the Infinity Cache, one kernel's steady-state clock -- do not rank tiles the way the denoise step does; as in rounds 3 / 4.)
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
    if (!p || p->struct_size != (uint32_t)sizeof(pcdm_gemm_params)) return -1;   // (a host built against another header: refused before any other field is read)
    if (!p->a || !p->w || !p->out) return -1;
    if (p->M <= 0 || p->N <= 0 || p->K <= 0 || p->K % BK || p->Npad % 64 || p->Npad < p->N || p->N % 4) return -1;
    if (p->rows_per_batch <= 0) return -1;
    // 16-byte vector loads of the epilogue operands: bias and rowvec bases, the row pitch and the per-step block stride
    if (((uintptr_t)p->bias & 15) || (p->rowvec && (((uintptr_t)p->rowvec & 15) || (p->ldrv > 0 && p->ldrv % 4) || (p->rowvec_step && p->rowvec_step_stride % 4))))
        return -1;
    // 32-bit buffer offsets: every operand must stay below 2 GiB
    const int64_t lim = 0x7fffffffLL;
    if ((int64_t)p->Npad * (p->ldw > 0 ? p->ldw : p->K) * 2 >= lim) return -2;
    if (p->conv ? ((int64_t)p->B * p->Hi * p->Wi * p->cin * 2 + ((int64_t)p->Wi + 1) * p->cin * 2 >= lim ||
                   (p->a2 && (int64_t)p->M * p->lda2 * 2 >= lim) || (p->a3 && (int64_t)p->M * p->lda3 * 2 >= lim))
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
    a.a3 = (const u16*)p->a3;
    a.lda3 = p->lda3;
    a.xinv = 0;
    a.tap_lut = 0;
    a.tap_group_n = 0;
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
    a.rowvec_step_count = a.rowvec_step ? p->rowvec_step_count : 0;
    a.step_error = a.rowvec_step ? p->step_error : nullptr;
    if (a.rowvec_step_count < 0 || ((uintptr_t)a.step_error & 3)) return -1;
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
    if (p->conv && p->tap_group_n > 0) {
        // a subset of the nine taps per output-channel group (pcdm_gemm_params.tap_lut): K = ntaps cin
        const int ntaps = p->cin > 0 ? p->K / p->cin : 0;
        if (p->cin <= 0 || p->cin % BK || p->K != ntaps * p->cin || ntaps < 1 || ntaps > 4 || p->stride != 1 || p->upsample || p->no_pad_lo || p->dup_rows ||
            p->a2 || p->a3 || p->Hi != p->Ho || p->Wi != p->Wo || p->N % p->tap_group_n || p->N / p->tap_group_n > 4 || p->tap_group_n % 64)
            return -1;
        for (int g = 0; g < p->N / p->tap_group_n; ++g)
            for (int t = 0; t < ntaps; ++t)
                if (((p->tap_lut >> (16 * g + 4 * t)) & 15) > 8 || ((p->tap_lut >> (16 * g)) & 0xffff) == 0) return -1;
        if (p->M != p->B * p->Ho * p->Wo) return -1;
        a.tap_lut = p->tap_lut;
        a.tap_group_n = p->tap_group_n;
    } else if (p->conv) {
        const int cx = p->K - 9 * p->cin;   // extra K behind the nine taps: a 1x1 convolution over a2 [+ a3] at the output pixel (pcdm_gemm_params.a3)
        if (p->cin <= 0 || p->cin % BK || cx < 0 || (p->stride != 1 && p->stride != 2)) return -1;
        if (cx == 0 ? (p->a2 || p->a3)
                    : (!p->a2 || p->stride != 1 || p->upsample || p->no_pad_lo || p->dup_rows || p->Hi != p->Ho || p->Wi != p->Wo || cx % BK ||
                       p->c1 <= 0 || p->c1 % BK || p->c1 > cx || (p->c1 < cx) != (p->a3 != nullptr) || p->lda2 < p->c1 || (p->lda2 & 7) ||
                       (p->a3 && (p->lda3 < cx - p->c1 || (p->lda3 & 7))) || (((uintptr_t)p->a2 | (uintptr_t)p->a3) & 15)))
            return -1;
        a.xinv = (uint32_t)(((1ull << 31) + (uint64_t)(p->cin / 64) - 1) / (uint64_t)(p->cin / 64));   // m = ((a_off >> 7) * xinv) >> 31: exact for a_off < 2^31
        if (p->upsample && (p->stride != 1 || p->no_pad_lo || p->Ho < p->Hi || p->Wo < p->Wi)) return -1;
        if (p->M != p->B * p->Ho * p->Wo) return -1;
    } else {
        if (p->a3 || (p->a2 && (p->c1 % BK || p->c1 <= 0 || p->c1 >= p->K))) return -1;
    }
    if (p->epilogue == PCDM_EPI_GEGLU && (!p->bias || p->Npad % 128 || p->N * 2 > p->Npad)) return -1;
    if (p->epilogue == PCDM_EPI_SPLIT_VT && (!p->out2 || p->vt_col0 % 4)) return -1;
    hipStream_t st = (hipStream_t)s;
    int tile = p->tile & 0xff;
    if (tile >= pcdm_gemm_detail::kRowGemmTile0) {
        if ((a.debug & 4) && p->ws_floats < (int64_t)((p->M + 95) / 96) * 8 * 8 * 2) return -1;   // (stamps: 8 x uint64 per wave)
        if (p->row_stats_out) return -1;   // (the A-in-registers kernel has no partials producer: refused, never silently ignored -- ADVICE r5)
        return p->conv ? -1 : pcdm_gemm_detail::launch_rowgemm(tile, a, st);
    }
    a.ln_row_stats = p->ln_row_stats;
    a.row_stats_out = p->row_stats_out;
    if (a.ln_wsum) {
        // the folded LayerNorm on a tiled instance (gemm_ext.hip): linear, single source, whole K in one workgroup, the LDS-staged epilogue only
        // (nothing that instance does not implement may be asked for: the generic epilogue knows nothing about the fold).  Row statistics:
        // from the producer's partials when given (ln_row_stats), else taken in the K loop
        if (p->conv || p->a2 || a.split_k > 1 || p->rowvec || p->residual || p->act || a.zero_rows || a.dup_rows || (p->N & 7) || (p->ldo & 7) ||
            ((uintptr_t)a.ln_wsum & 15) || p->row_stats_out)
            return -1;
        if (p->epilogue != PCDM_EPI_STORE && p->epilogue != PCDM_EPI_GEGLU && p->epilogue != PCDM_EPI_SPLIT_VT) return -1;
        if (p->epilogue == PCDM_EPI_SPLIT_VT && (p->vt_col0 % 64 || p->rows_per_batch % 32 || (p->ldo2 & 7) || p->M % 32)) return -1;
        if (p->epilogue == PCDM_EPI_GEGLU && (tile == 2 || tile == 8)) return -1;   // (GEGLU pairs need a 64-wide wave tile)
        if (((tile == 4 || tile == 7 || tile == 18) && p->Npad % 128) || ((tile == 17 || tile == 26 || tile == 23) && p->Npad % 256)) return -1;
        if (a.ln_row_stats && ((p->K & 63) || p->K > 1280 || ((uintptr_t)a.ln_row_stats & 15))) return -1;   // (16-byte loads of pair couples; <= 40 pairs per row)
        return pcdm_gemm_detail::launch_gemm_ext(a.ln_row_stats ? 2 : 1, tile, a, st);
    }
    if (a.ln_row_stats) return -1;
    if (a.row_stats_out) {
        // row-statistics producer (gemm_ext.hip): a linear STORE launch whose lean epilogue also writes the {sum, M2} of every 32-column run of
        // the rows it stores
        if (p->conv || a.split_k > 1 || p->act || a.dup_rows || p->epilogue != PCDM_EPI_STORE || (p->N & 31) || (p->ldo & 7) || ((uintptr_t)a.row_stats_out & 7) ||
            (p->residual && ((p->ldr & 7) || (p->res_mod > 0 && p->res_mod < p->M))) || (p->rowvec && p->rows_per_batch < 32))
            return -1;
        if ((tile == 4 || tile == 7 || tile == 18) && p->Npad % 128) return -1;
        return pcdm_gemm_detail::launch_gemm_ext(3, tile, a, st);
    }
    // (a tap-subset convolution needs N tiles that lie inside one output-channel group: the heuristic's 128-wide tiles only when the group allows)
    const bool n128 = p->Npad % 128 == 0 && (a.tap_group_n == 0 || a.tap_group_n % 128 == 0);
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

