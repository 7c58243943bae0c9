// Micro-benchmark (not part of the library): what do v_mfma_f32_32x32x16_bf16 loops sustain on this MI355X
//   mode 0: MFMA only (operands in registers)
//   mode 1: MFMA fed by swizzled ds_read_b128 fragments from a resident LDS tile, no barriers, no global loads
// for wave tiles FMxFN of 32x32 fragments and W waves per workgroup.   hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int FM, int FN, int MODE>
__global__ void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u16* As = (u16*)smem;  // [256][64] swizzled
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 256 * 64; i += blockDim.x) As[i] = (u16)(0x3f80 + (i & 7));
    __syncthreads();
    f32x16 acc[FN][FM];
    for (int i = 0; i < FN; ++i)
        for (int j = 0; j < FM; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fsw = (lane >> 1) & 7, fhalf = lane >> 5;
    const u16* as = As + ((wave & 1) * 64 + frow) * 64;
    const u16* bs = As + (128 + (wave & 1) * 32 + frow) * 64;
    u16x8 xf[FM], wf[FN];
    for (int j = 0; j < FM; ++j) xf[j] = *(const u16x8*)(as + (j & 1) * 32 * 64);
    for (int i = 0; i < FN; ++i) wf[i] = *(const u16x8*)(bs + (i & 1) * 32 * 64);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2 || (MODE == 3 && (it & 1) == 0)) asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (MODE >= 1) {
                const int co = ((ks * 2 + fhalf) ^ fsw) * 8;
#pragma unroll
                for (int j = 0; j < FM; ++j) xf[j] = *(const u16x8*)(as + (j & 1) * 32 * 64 + (j >> 1) * 2048 + co);
#pragma unroll
                for (int i = 0; i < FN; ++i) wf[i] = *(const u16x8*)(bs + (i & 1) * 32 * 64 + (i >> 1) * 2048 + co);
            }
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[i]),
                                                                       __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < FN; ++i)
        for (int j = 0; j < FM; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FM, int FN, int MODE>
void run(int waves, int blocks_per_cu, const char* name) {
    float* out;
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    hipMalloc(&out, (size_t)blocks * waves * 64 * 4);
    hipFuncSetAttribute((const void*)k<FM, FN, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<FM, FN, MODE><<<blocks, waves * 64, 32768>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<FM, FN, MODE><<<blocks, waves * 64, 32768>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * waves * iters * 4 * FM * FN * 32768.0;
    printf("%-34s waves/blk %d blk/CU %d : %8.1f TF/s  (%.3f ms)\n", name, waves, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    run<2, 2, 0>(4, 1, "mfma only 64x64 tile");
    run<2, 2, 0>(4, 2, "mfma only 64x64 tile");
    run<2, 2, 0>(8, 1, "mfma only 64x64 tile");
    run<2, 2, 1>(4, 1, "lds-fed 64x64 (1 KB/MFMA)");
    run<2, 2, 1>(4, 2, "lds-fed 64x64 (1 KB/MFMA)");
    run<2, 2, 1>(8, 1, "lds-fed 64x64 (1 KB/MFMA)");
    run<2, 2, 1>(8, 2, "lds-fed 64x64 (1 KB/MFMA)");
    run<2, 2, 2>(4, 1, "lds-fed 64x64 + barrier/16 MFMA");
    run<2, 2, 2>(4, 2, "lds-fed 64x64 + barrier/16 MFMA");
    run<2, 2, 2>(8, 1, "lds-fed 64x64 + barrier/16 MFMA");
    run<2, 2, 2>(8, 2, "lds-fed 64x64 + barrier/16 MFMA");
    run<2, 2, 3>(4, 2, "lds-fed 64x64 + barrier/32 MFMA");
    run<2, 2, 3>(8, 1, "lds-fed 64x64 + barrier/32 MFMA");
    run<2, 1, 1>(8, 2, "lds-fed 64x32 (1.5 KB/MFMA)");
    run<1, 1, 1>(4, 4, "lds-fed 32x32 (2 KB/MFMA)");
    return 0;
}
