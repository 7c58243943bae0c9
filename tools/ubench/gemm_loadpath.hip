// Micro-benchmark (not part of the library): cost of the global -> LDS staging path inside an MFMA main loop on MI355X.
// A 128x128x64 (4 waves) or 256x128x64 (8 waves) block tile per iteration: 16 MFMAs per wave fed by swizzled ds_read_b128
// fragments from a 2-stage LDS ring, one counted wait + raw barrier per K-tile, tile t+1 fetched while tile t is computed.
//   mode 0: no global traffic (LDS-resident data)                       -- the ceiling
//   mode 1: LDS-DMA (buffer_load_dwordx4 ... lds), what gemm.hip does
//   mode 2: register-staged (buffer_load_dwordx4 -> VGPRs, later ds_write_b128)
//   mode 3 / 4: LDS-DMA pieces spread over the four k-steps, issued after / before each k-step's MFMAs
//   mode 5: LDS-DMA, 3-stage ring, two tiles in flight, counted s_waitcnt
//   KT = 2: BK = 128 (two 64-deep K-tiles per stage and per barrier)
// The source (8 MiB, L2 / MALL resident) is read with per-lane constant offsets + an SGPR offset, like the real kernel.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o gemm_loadpath gemm_loadpath.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Rsrc { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ Rsrc make_rsrc(const void* p) { return Rsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000)}; }
__device__ __forceinline__ void dma16(Rsrc r, uint32_t voff, uint32_t soff, void* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r.r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ u32x4 load16(Rsrc r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b128(r.r, voff, soff, 0); }
__device__ __forceinline__ f32x16 mfma(u16x8 a, u16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void wait_all_then_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_vm_then_barrier() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int NW, int MODE, int KT = 1>
__global__ __launch_bounds__(NW * 64) void k(const u16* __restrict__ src, uint32_t src_bytes, float* out, int iters) {
    constexpr int ROWS = NW == 4 ? 256 : 384;   // A rows + B rows of the block tile (128+128 or 256+128)
    constexpr int PIECES = KT * ROWS / 8 / NW;  // 1 KiB pieces per wave per stage (8 rows x 128 B each; KT K-tiles of 64 per stage)
    constexpr int STAGE = KT * ROWS * 64;       // u16 elements per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u16* ring = (u16*)smem;                     // [2 or 3][ROWS][64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * STAGE; i += blockDim.x) ring[i] = (u16)(0x3f80 + (i & 7));
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fsw = (lane >> 1) & 7, fhalf = lane >> 5;
    const int arow = (NW == 4 ? (wave >> 1) * 64 : (wave >> 1) * 64) + frow;          // wave grid (NW/2) x 2, 64x64 wave tiles
    const int brow = (NW == 4 ? 128 : 256) + (wave & 1) * 64 + frow;
    const Rsrc rs = make_rsrc(src);
    uint32_t voff[PIECES];
    for (int j = 0; j < PIECES; ++j) voff[j] = (uint32_t)(((blockIdx.x * 37 + wave * PIECES + j) * 1024) % (src_bytes / 2)) + lane * 16;
    u32x4 regs[PIECES];
    auto issue = [&](int it, int stage) {
        const uint32_t soff = (uint32_t)((it * 8192) % (src_bytes / 2));
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j)
                dma16(rs, voff[j], soff, ring + stage * STAGE + (wave * PIECES + j) * 512);
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j) regs[j] = load16(rs, voff[j], soff);
        }
    };
    auto issue_piece = [&](int it, int stage, int j) {
        const uint32_t soff = (uint32_t)((it * 8192) % (src_bytes / 2));
        dma16(rs, voff[j], soff, ring + stage * STAGE + (wave * PIECES + j) * 512);
    };
    constexpr int NST = MODE == 5 ? 3 : 2;
    issue(0, 0);
    if (MODE == 5) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) issue_piece(0, 0, j);
#pragma unroll
        for (int j = 0; j < PIECES; ++j) issue_piece(1, 1, j);
    }
    if (MODE == 3 || MODE == 4) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) issue_piece(0, 0, j);
    }
    if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) *(u32x4*)(ring + (wave * PIECES + j) * 512 + lane * 8) = regs[j];
    }
    int cur = 0, nxt = MODE == 5 ? 2 : 1;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 5) wait_vm_then_barrier<PIECES>();   // tile `it` landed; tile it+1 stays in flight
        else wait_all_then_barrier();                    // tile `it` is in LDS for everybody
        if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j) issue_piece(it + 2, nxt, j);
        } else if (MODE != 3 && MODE != 4) {
            issue(it + 1, nxt);
        }
#pragma unroll
        for (int ks4 = 0; ks4 < 4 * KT; ++ks4) {
            const int ks = ks4 & 3;
            const u16* as = ring + cur * STAGE + (ks4 >> 2) * ROWS * 64 + arow * 64;
            const u16* bs = ring + cur * STAGE + (ks4 >> 2) * ROWS * 64 + brow * 64;
            const int co = ((ks * 2 + fhalf) ^ fsw) * 8;
            u16x8 xf[2], wf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) xf[j] = *(const u16x8*)(as + j * 32 * 64 + co);
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[i] = *(const u16x8*)(bs + i * 32 * 64 + co);
            if (MODE == 4) {
#pragma unroll
                for (int j = ks4 * PIECES / (4 * KT); j < (ks4 + 1) * PIECES / (4 * KT); ++j) issue_piece(it + 1, nxt, j);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma(wf[i], xf[j], acc[i][j]);
            if (MODE == 3) {
#pragma unroll
                for (int j = ks4 * PIECES / (4 * KT); j < (ks4 + 1) * PIECES / (4 * KT); ++j) issue_piece(it + 1, nxt, j);
            }
        }
        if (MODE == 2) {   // the fetched tile goes to the other stage (its readers finished before the barrier above)
            wait_vm0();
#pragma unroll
            for (int j = 0; j < PIECES; ++j) *(u32x4*)(ring + nxt * STAGE + (wave * PIECES + j) * 512 + lane * 8) = regs[j];
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NW, int MODE, int KT = 1>
void run(int blocks_per_cu, const u16* src, uint32_t src_bytes, const char* name) {
    float* out;
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    constexpr int smem = (MODE == 5 ? 3 : 2) * KT * (NW == 4 ? 256 : 384) * 128;
    hipMalloc(&out, (size_t)blocks * NW * 64 * 4);
    hipFuncSetAttribute((const void*)k<NW, MODE, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NW, MODE, KT><<<blocks, NW * 64, smem>>>(src, src_bytes, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NW, MODE, KT><<<blocks, NW * 64, smem>>>(src, src_bytes, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * NW * iters * 16 * KT * 32768.0;
    printf("%-58s blk/CU %d : %8.1f TF/s  (%.3f ms)\n", name, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    const uint32_t bytes = 8u << 20;
    u16* src;
    hipMalloc(&src, bytes);
    hipMemset(src, 0x3f, bytes);
    run<4, 0>(2, src, bytes, "128x128/4 waves, LDS-resident (ceiling)");
    run<4, 1>(2, src, bytes, "128x128/4 waves, LDS-DMA 8 x 1 KiB per wave per K-tile");
    run<4, 2>(2, src, bytes, "128x128/4 waves, register-staged (8 loads + 8 ds_write_b128)");
    run<4, 3>(2, src, bytes, "128x128/4 waves, LDS-DMA spread over k-steps, after MFMAs");
    run<4, 4>(2, src, bytes, "128x128/4 waves, LDS-DMA spread over k-steps, before MFMAs");
    run<4, 5>(1, src, bytes, "128x128/4 waves, LDS-DMA 3 stages / 2 tiles in flight");
    run<8, 0>(1, src, bytes, "256x128/8 waves, LDS-resident (ceiling)");
    run<8, 1>(1, src, bytes, "256x128/8 waves, LDS-DMA 6 x 1 KiB per wave per K-tile");
    run<8, 2>(1, src, bytes, "256x128/8 waves, register-staged (6 loads + 6 ds_write_b128)");
    run<4, 1, 2>(1, src, bytes, "128x128/4 waves, LDS-DMA, BK = 128 (one barrier per 32 MFMAs)");
    run<4, 3, 2>(1, src, bytes, "128x128/4 waves, LDS-DMA spread, BK = 128");
    run<8, 3>(1, src, bytes, "256x128/8 waves, LDS-DMA spread over k-steps, after MFMAs");
    run<8, 5>(1, src, bytes, "256x128/8 waves, LDS-DMA 3 stages / 2 tiles in flight");
    return 0;
}
