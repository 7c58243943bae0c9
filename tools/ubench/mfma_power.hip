// Micro-benchmark (not part of the library), round 6: the PRACTICAL matrix-pipe roof of this MI355X.
//
//   A  register-resident MFMA-only loops (v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16; no LDS, no memory traffic in the loop) on
//      zero / constant / N(0,1) operands, at one and two waves per SIMD.  The chip clocks to its power budget (MI355X_MICROARCH.md "DVFS
//      give-back"): with N(0,1) operands the same instruction stream runs at a lower clock than with zeros.  The N(0,1) number is the roof
//      every MFMA fraction of DESIGN.md is ALSO quoted against from round 6 on (next to the 2.5 PF/s datasheet figure).
//   B  the K loop of a 176 x 320 block tile on FOUR waves (one per SIMD, 176 x 80 wave tiles = 11 x 5 fragments of 16x16x32, accumulators in
//      AGPRs, 512 registers per lane), fed from a resident two-stage LDS tile with the library's chunk swizzle and one barrier per 64-deep
//      K-tile -- no global loads: the "no-load loop" of tools/ablate_gemm.py for a tile that does not exist in the library yet.  Kill criterion
//      of the round-5 review: < 1.35 PF/s on N(0,1) operands (the 8-wave 192 x 320 tile's no-load loop: 1.12-1.29).
//   C  the same loop for the 8-wave 96 x 80 wave tile (2 waves per SIMD, VGPR accumulators) as the control.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static inline u16 f2bf_host(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (u16)(u >> 16);
}

// ---------------------------------------------------------------------------------------------------------- A: registers only
// FRAG 16: FA x FB accumulators of 4 registers; FRAG 32: of 16.  Operands: FA + FB fragments per lane, loaded once.
template <int FRAG, int FA, int FB, int WPS>
__global__ __launch_bounds__(256 * WPS) void mfma_regs(const u16* __restrict__ src, float* out, long long* clk, int iters) {
    const int t = threadIdx.x;
    u16x8 a[FA], b[FB];
    for (int j = 0; j < FA; ++j) a[j] = *(const u16x8*)(src + ((size_t)(blockIdx.x * 7 + j) * 1024 + t) % (1 << 20) * 8);
    for (int i = 0; i < FB; ++i) b[i] = *(const u16x8*)(src + ((size_t)(blockIdx.x * 13 + 64 + i) * 1024 + t) % (1 << 20) * 8);
    typedef typename std::conditional<FRAG == 16, f32x4, f32x16>::type acc_t;
    acc_t acc[FB][FA];
    for (int i = 0; i < FB; ++i)
        for (int j = 0; j < FA; ++j)
            for (int r = 0; r < (FRAG == 16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < FB; ++i)
#pragma unroll
            for (int j = 0; j < FA; ++j) {
                if constexpr (FRAG == 16)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[i]), __builtin_bit_cast(bf16x8, a[j]), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[i]), __builtin_bit_cast(bf16x8, a[j]), acc[i][j], 0, 0, 0);
            }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < FB; ++i)
        for (int j = 0; j < FA; ++j)
            for (int r = 0; r < (FRAG == 16 ? 4 : 16); ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * blockDim.x + t] = s;
    if (t == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

// ---------------------------------------------------------------------------------------------------------- B / C: LDS-fed K loop
// Block tile BM x 320 (BM = 176 on 4 waves: FM = 11; BM = 192 on 8 waves: FM = 6, two wave rows), K-tile 64 = two k-steps of 32, two stages
// resident in LDS (filled once from `src`), chunk swizzle (row >> 1) & 7 as in gemm_kernel.inc.  Per k-step a wave reads FM + 5 fragments
// (ds_read_b128) for the NEXT k-step while the FM x 5 MFMAs of the current one issue; the K-tile barrier sits in front of the last k-step's
// MFMAs ("rotated").  No global memory traffic inside the loop.
template <int FM, int WGM, int MPR /* MFMAs between two ds_reads in the schedule */>
__global__ __launch_bounds__(WGM * 4 * 64) void kloop_lds(const u16* __restrict__ src, float* out, long long* clk, int nkt) {
    constexpr int FN = 5, BK = 64, BN = 320, BMS = 192, STAGES = 2, F = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u16* As = (u16*)smem;                     // [2][192][64]
    u16* Bs = As + STAGES * BMS * BK;         // [2][320][64]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / 4, wn = wave & 3;
    for (int i = t; i < STAGES * (BMS + BN) * BK / 8; i += blockDim.x)
        ((u16x8*)smem)[i] = *(const u16x8*)(src + ((size_t)blockIdx.x * 4099 + i) % (1 << 20) * 8);
    __syncthreads();
    f32x4 acc[FN][FM];
    for (int i = 0; i < FN; ++i)
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fsw = (frow >> 1) & 7, fhalf = lane >> 4;
    u16x8 xf[2][FM], wf[2][FN];
    auto load_frags = [&](int stage, int ks, auto bsel) {
        constexpr int b = decltype(bsel)::value;
        const int co = ((ks * 4 + fhalf) ^ fsw) * 8;
        const u16* as = As + stage * BMS * BK + (wm * FM * F + frow) * BK + co;
        const u16* bs = Bs + stage * BN * BK + (wn * 80 + frow) * BK + co;
#pragma unroll
        for (int j = 0; j < FM; ++j) xf[b][j] = *(const u16x8*)(as + j * F * BK);
#pragma unroll
        for (int i = 0; i < FN; ++i) wf[b][i] = *(const u16x8*)(bs + i * F * BK);
    };
    auto mfma_block = [&](auto bsel) {
        constexpr int b = decltype(bsel)::value;
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[b][i]), __builtin_bit_cast(bf16x8, xf[b][j]), acc[i][j], 0, 0, 0);
    };
    // one scheduling region = (FM + FN) ds_reads of the next k-step + FM * FN MFMAs of this one, interleaved MPR MFMAs : 1 read
    auto interleave = [&]() {
        if constexpr (MPR > 0) {
#pragma unroll
            for (int r = 0; r < FM + FN; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, FM * FN - MPR * (FM + FN), 0);
        }
    };
    typedef std::integral_constant<int, 0> B0;
    typedef std::integral_constant<int, 1> B1;
    load_frags(0, 0, B0());
    const long long c0 = clock64(), w0 = wall_clock64();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        load_frags(cur, 1, B1());
        mfma_block(B0());
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        // every wave has read both k-steps of this stage: barrier (in the library: the stage is refilled by LDS-DMA right behind it)
        __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nx = cur ^ 1;
        load_frags(nx, 0, B0());
        mfma_block(B1());
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        cur = nx;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < FN; ++i)
        for (int j = 0; j < FM; ++j)
            for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * blockDim.x + t] = s;
    if (t == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

static u16* g_src[3];
static float* g_out;
static long long* g_clk;
static const char* kData[3] = {"zeros", "ones", "N(0,1)"};

static void report(const char* name, const char* data, double flops, float ms, int blocks) {
    std::vector<long long> c(2 * blocks);
    hipMemcpy(c.data(), g_clk, c.size() * 8, hipMemcpyDeviceToHost);
    double cs = 0, ws = 0;
    for (int i = 0; i < blocks; ++i) { cs += (double)c[2 * i]; ws += (double)c[2 * i + 1]; }
    // wall_clock64 ticks at 100 MHz; clock64 = s_memtime
    printf("%-58s %-7s %8.1f TF/s  (%.3f ms)  s_memtime/wall_clock = %.3f (x 100 MHz = %.0f MHz)\n", name, data, flops / ms / 1e9, ms, cs / ws,
           cs / ws * 100.0);
}

template <int FRAG, int FA, int FB, int WPS>
static void run_regs(const char* name) {
    const int blocks = 256, iters = 40000 / (FA * FB) * (FRAG == 16 ? 2 : 1);
    for (int d = 0; d < 3; ++d) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        mfma_regs<FRAG, FA, FB, WPS><<<blocks, 256 * WPS>>>(g_src[d], g_out, g_clk, iters / 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mfma_regs<FRAG, FA, FB, WPS><<<blocks, 256 * WPS>>>(g_src[d], g_out, g_clk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * WPS * iters * FA * FB * 2.0 * FRAG * FRAG * (512 / FRAG);
        report(name, kData[d], flops, ms, blocks);
    }
}

template <int FM, int WGM, int MPR>
static void run_kloop(const char* name) {
    const int blocks = 256, nkt = 2000;
    const int smem = 2 * (192 + 320) * 64 * 2;
    hipFuncSetAttribute((const void*)kloop_lds<FM, WGM, MPR>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    for (int d = 0; d < 3; ++d) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        kloop_lds<FM, WGM, MPR><<<blocks, WGM * 256, smem>>>(g_src[d], g_out, g_clk, nkt / 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kloop_lds<FM, WGM, MPR><<<blocks, WGM * 256, smem>>>(g_src[d], g_out, g_clk, nkt);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * nkt * 2.0 * (FM * 16 * WGM) * 320 * 64;
        report(name, kData[d], flops, ms, blocks);
    }
}

int main() {
    const size_t n = (size_t)8 << 20;   // bf16 elements per operand pool
    std::vector<u16> h(n);
    for (int d = 0; d < 3; ++d) {
        hipMalloc(&g_src[d], n * 2);
        if (d == 0) for (size_t i = 0; i < n; ++i) h[i] = 0;
        if (d == 1) for (size_t i = 0; i < n; ++i) h[i] = 0x3f80;
        if (d == 2) {
            srand(1234);
            for (size_t i = 0; i < n; i += 2) {
                const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
                const double r = sqrt(-2.0 * log(u1));
                h[i] = f2bf_host((float)(r * cos(6.283185307179586 * u2)));
                h[i + 1] = f2bf_host((float)(r * sin(6.283185307179586 * u2)));
            }
        }
        hipMemcpy(g_src[d], h.data(), n * 2, hipMemcpyHostToDevice);
    }
    hipMalloc(&g_out, (size_t)256 * 1024 * 4);
    hipMalloc(&g_clk, 256 * 2 * 8);
    printf("== A: register-resident MFMA only (256 workgroups, one per CU)\n");
    run_regs<32, 2, 2, 1>("32x32x16  4 acc   1 wave/SIMD");
    run_regs<32, 2, 2, 2>("32x32x16  4 acc   2 waves/SIMD");
    run_regs<32, 4, 3, 1>("32x32x16 12 acc   1 wave/SIMD");
    run_regs<16, 4, 4, 1>("16x16x32 16 acc   1 wave/SIMD");
    run_regs<16, 4, 4, 2>("16x16x32 16 acc   2 waves/SIMD");
    run_regs<16, 6, 5, 2>("16x16x32 30 acc   2 waves/SIMD (the 96x80 wave tile)");
    run_regs<16, 11, 5, 1>("16x16x32 55 acc   1 wave/SIMD (a 176x80 wave tile)");
    printf("== B / C: LDS-fed K loop, two resident stages, one barrier per K-tile, no global loads\n");
    run_kloop<6, 2, 0>("C  8 waves x 96x80   compiler schedule");
    run_kloop<6, 2, 2>("C  8 waves x 96x80   2 MFMA : 1 ds_read");
    run_kloop<11, 1, 0>("B  4 waves x 176x80  compiler schedule");
    run_kloop<11, 1, 3>("B  4 waves x 176x80  3 MFMA : 1 ds_read");
    run_kloop<11, 1, 2>("B  4 waves x 176x80  2 MFMA : 1 ds_read");
    return 0;
}
