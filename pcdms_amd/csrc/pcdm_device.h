// Device-side helpers shared by every pcdm kernel (gfx950 / CDNA4, wave = 64).
//
// PCDM_EMU is defined ONLY by the test build (tests/emu): the same kernel sources are then
// compiled for the host against a lane-level emulator so index logic can be checked on the
// GPU-less dev container.  The product library is always built by hipcc for gfx950.
#pragma once
#include <stdint.h>

#ifndef PCDM_EMU
#include <hip/hip_runtime.h>
#endif

typedef unsigned short u16;
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

#define PCDM_WAVE 64

// ---- bf16 <-> f32 (bit exact, round-to-nearest-even; NaN preserved) ------------------------
__device__ __forceinline__ float bf2f(u16 v) {
    uint32_t u = ((uint32_t)v) << 16;
    return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ u16 f2bf(float f) {
#ifndef PCDM_EMU
    // gfx950 has a hardware RNE convert (v_cvt_pk_bf16_f32); hipcc emits it for the native __bf16 cast
    return __builtin_bit_cast(u16, (__bf16)f);
#endif
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}

// two fp32 -> one dword of two bf16 (lo in bits 0..15): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
#ifndef PCDM_EMU
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
#else
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
#endif
}

// ---- launch / dynamic LDS ------------------------------------------------------------------
#ifdef PCDM_EMU
#define PCDM_DYN_SMEM(name) char* name = emu::dyn_smem
#define PCDM_LAUNCH(kernel, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define PCDM_KERNEL_NAME(...) __VA_ARGS__
#else
#define PCDM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define PCDM_KERNEL_NAME(...) __VA_ARGS__
// hipGetLastError() first: drop any stale error left by an unrelated runtime call on this thread, so the
// PCDM_CHECK_LAUNCH() after the launch reports this launch only
#define PCDM_LAUNCH(kernel, grid, block, smem, stream, ...) \
    ((void)hipGetLastError(), kernel<<<(grid), (block), (smem), (stream)>>>(__VA_ARGS__))
#endif

// ---- async global -> LDS copy (LDS-DMA, global_load_lds_dwordx4) -----------------------------
// Each lane supplies its own 16-byte GLOBAL source; the destination is wave-uniform:
// lane i lands at lds_wave_base + 16*i.  Completion is tracked by vmcnt: call glds_wait() before the
// barrier that publishes the tile.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
#ifdef PCDM_EMU
    memcpy((char*)lds_wave_base + 16 * emu::lane_id(), gsrc, 16);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}
__device__ __forceinline__ void glds_wait() {
#ifndef PCDM_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ---- LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... offen lds) ------------------------
// address = base + voff (per lane, VGPR) + soff (wave-uniform, SGPR); lanes whose voff has bit 31 set are out
// of range: the hardware bounds check makes them deliver ZEROS (no branch, no separate zero source).
// an optimisation barrier on one VGPR value: what is computed from it afterwards cannot be hoisted out of the enclosing loop (rarely taken
// branches of a register-bound loop: the loop-invariant code motion of their address arithmetic would cost the common path live registers)
#ifdef PCDM_EMU
#define PCDM_LOOP_VARIANT(x) ((void)0)
#else
#define PCDM_LOOP_VARIANT(x) asm volatile("" : "+v"(x))
#endif
#ifdef PCDM_EMU
struct BufRsrc { const char* base; uint32_t size; };
__device__ __forceinline__ BufRsrc make_buf_rsrc(const void* p, uint32_t bytes = 0x7fffffffu) { return BufRsrc{(const char*)p, bytes}; }
__device__ __forceinline__ void buf_glds16(BufRsrc r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    char* d = (char*)lds_wave_base + 16 * emu::lane_id();
    if (voff >= r.size) memset(d, 0, 16);
    else memcpy(d, r.base + voff + soff, 16);
}
// 16-byte register load / store through a descriptor: voff >= size (the bounds check is on the per-lane offset) reads zeros /
// drops the store -- rows beyond the tensor and masked lanes (voff bit 31) need no predicate and no branch
__device__ __forceinline__ u32x4 buf_load16(BufRsrc r, uint32_t voff) {
    u32x4 v = {0, 0, 0, 0};
    if (voff < r.size && voff + 16 <= r.size) memcpy(&v, r.base + voff, 16);
    return v;
}
__device__ __forceinline__ void buf_store16(BufRsrc r, uint32_t voff, u32x4 v) {
    if (voff < r.size && voff + 16 <= r.size) memcpy(const_cast<char*>(r.base) + voff, &v, 16);
}
__device__ __forceinline__ u32x4 buf_load16_once(BufRsrc r, uint32_t voff) { return buf_load16(r, voff); }
#else
struct BufRsrc { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ BufRsrc make_buf_rsrc(const void* p, uint32_t bytes = 0x7fffffffu) {
    return BufRsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000)};
}
__device__ __forceinline__ void buf_glds16(BufRsrc r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r.r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
// 16-byte register load / store through a descriptor: voff >= num_records (the bounds check is on the per-lane offset) reads zeros /
// drops the store -- rows beyond the tensor and masked lanes (voff bit 31) need no predicate and no branch
__device__ __forceinline__ u32x4 buf_load16(BufRsrc r, uint32_t voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r.r, voff, 0, 0));
}
// residual rows of an epilogue: read exactly once per launch -> nt (streaming: they do not displace the operand tiles in the L2).  Same-box A/B
// +0.15 % in both interleaved pairs (profiles/r5_ab_load_policy.json: at the edge of the 0.1 % repeatability; the same hint on the GroupNorm inputs
// was 0.35 % slower and is not used).  PCDM_BUILD_DEFINES="PCDM_RESLOAD_AUX=0": plain loads.
#ifndef PCDM_RESLOAD_AUX
#define PCDM_RESLOAD_AUX 2
#endif
__device__ __forceinline__ u32x4 buf_load16_once(BufRsrc r, uint32_t voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r.r, voff, 0, PCDM_RESLOAD_AUX));
}
// PCDM_STORE_AUX: cache-policy bits of the kernels' OUTPUT stores (gfx942 / gfx950: 1 = sc0, 2 = nt, 16 = sc1).  Round 5: **16 (sc1, agent scope) is
// the default**: an agent-scope store is written through the XCD's private L2 while the kernel runs, so the launch does not end with a burst
// write-back of up to 32 MB of dirty lines in front of the next (dependent) launch -- which has to wait for exactly that write-back, since the
// eight L2s are not coherent with each other.  Same-box A/B of the GEMM / rowgemm epilogue stores alone (profiles/r5_ab_store_scope.json):
// sc1 +0.7 % / +0.4 % (two boxes), sc0 sc1 +0.7 %, sc0 -0.1 %, nt (round 2) -1.8 %.  NOT for the norms' / the attention's outputs and the split-K
// slabs: with those written through as well the step was 1.2 % SLOWER (norm outputs alone -0.3 %: their consumer, a 3x3 convolution, reads its
// input nine times and found the lines in the L2 before; attention outputs alone -1.1 %: 8-byte stores per lane).  PCDM_BUILD_DEFINES="PCDM_STORE_AUX=0": the write-back stores of rounds 1-4.
#ifndef PCDM_STORE_AUX
#define PCDM_STORE_AUX 16
#endif
__device__ __forceinline__ void buf_store16(BufRsrc r, uint32_t voff, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r.r, voff, 0, PCDM_STORE_AUX);
}
#endif

#ifdef PCDM_EMU
#define PCDM_SCHED_BARRIER() ((void)0)
#define PCDM_SETPRIO(n) ((void)0)
// lanes of a wave run in lockstep on the GPU; the emulator's fibers need an explicit rendezvous
#define PCDM_WAVE_SYNC() ((void)__shfl_xor(0, 1, 64))
#else
#define PCDM_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#define PCDM_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#define PCDM_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

// ---- math ----------------------------------------------------------------------------------
__device__ __forceinline__ float fast_exp2(float x) {
#ifdef PCDM_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);  // v_exp_f32
#endif
}
__device__ __forceinline__ float fast_rcp(float x) {
#ifdef PCDM_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf-based GELU (PyTorch F.gelu default, diffusers GEGLU):  gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|).
// Phi(-t) = 2^Q(t) with Q a degree-5 minimax fit of log2(Phi(-t)) on [0, 6] (weighted by t Phi(-t), i.e. minimising the ABSOLUTE error
// of the result; tools/fit_gelu.py derives and checks the coefficients); |x| is clamped to 6, where the term is 6e-9.
// |gelu_erf_f(x) - gelu(x)| <= 7e-7 for every x (fp32 evaluation included), <= 2^-11 relative wherever |gelu(x)| >= 2e-3 -- a quarter of
// the bf16 half-ulp of the result.  ONE transcendental (v_exp_f32) and FMAs: the Abramowitz-Stegun 7.1.26 form it replaces (v_rcp +
// v_exp + 5-term Horner + sign select, same absolute error) was 40 % of the level-0 GEGLU launch (profiles/r3_rowgemm_anatomy.txt);
// quarter-rate instructions cost four issue slots each.  The two-wide form lets hipcc emit v_pk_fma_f32 / v_pk_mul_f32 (two lanes'
// worth of fp32 per issue slot) for the polynomial.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PCDM_GELU_Q0 (-1.000037670135498f)
#define PCDM_GELU_Q1 (-1.1507878303527832f)
#define PCDM_GELU_Q2 (-0.4599924385547638f)
#define PCDM_GELU_Q3 (-0.051827382296323776f)
#define PCDM_GELU_Q4 (0.007084557320922613f)
#define PCDM_GELU_Q5 (-0.0004733092791866511f)
#ifdef PCDM_GELU_AS7126   // (A/B build, tools/README.md: the Abramowitz-Stegun form of rounds 1-3)
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = fast_rcp(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * fast_exp2(-1.44269504088896341f * z * z);   // erf(|x|/sqrt2)
    return 0.5f * x * (1.0f + (x < 0.f ? -e : e));
}
__device__ __forceinline__ f32x2 gelu_erf_f2(f32x2 x) { return f32x2{gelu_erf_f(x[0]), gelu_erf_f(x[1])}; }
#else
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float t = fminf(fabsf(x), 6.0f);
    const float q = PCDM_GELU_Q0 + t * (PCDM_GELU_Q1 + t * (PCDM_GELU_Q2 + t * (PCDM_GELU_Q3 + t * (PCDM_GELU_Q4 + t * PCDM_GELU_Q5))));
    // (x > 0 ? x : x != x ? x : 0: fmaxf / fminf drop a NaN operand, which laundered an overflowed gate into a finite output -- ADVICE r4)
    return (x > 0.f ? x : (x != x ? x : 0.f)) - t * fast_exp2(q);
}
__device__ __forceinline__ f32x2 gelu_erf_f2(f32x2 x) {
    const f32x2 t = {fminf(fabsf(x[0]), 6.0f), fminf(fabsf(x[1]), 6.0f)};
    f32x2 q = t * PCDM_GELU_Q5 + PCDM_GELU_Q4;
    q = q * t + PCDM_GELU_Q3;
    q = q * t + PCDM_GELU_Q2;
    q = q * t + PCDM_GELU_Q1;
    q = q * t + PCDM_GELU_Q0;
    const f32x2 e = {fast_exp2(q[0]), fast_exp2(q[1])};
    const f32x2 r = {x[0] > 0.f ? x[0] : (x[0] != x[0] ? x[0] : 0.f), x[1] > 0.f ? x[1] : (x[1] != x[1] ? x[1] : 0.f)};   // NaN-propagating max(x, 0)
    return r - t * e;
}
#endif
// v[e] = h[e] * gelu(g[e]), e = 0..3 (one accumulator quad of the GEGLU epilogues)
__device__ __forceinline__ f32x4 geglu_quad(f32x4 h, f32x4 g) {
    const f32x2 a = gelu_erf_f2(f32x2{g[0], g[1]}), b = gelu_erf_f2(f32x2{g[2], g[3]});
    const f32x2 ha = {h[0], h[1]}, hb = {h[2], h[3]};
    const f32x2 va = ha * a, vb = hb * b;
    return f32x4{va[0], va[1], vb[0], vb[1]};
}

// ---- MFMA (cdna_hip_programming.md §3 fragment maps) -----------------------------------------
// 32x32x16 bf16:  A lane l -> row  (l&31), k = 8*(l>>5)+e ;  B lane l -> col (l&31), same k ;
//                 D lane l, reg r -> col (l&31), row (r&3) + 8*(r>>2) + 4*(l>>5).
__device__ __forceinline__ f32x16 mfma_32x32x16(u16x8 a, u16x8 b, f32x16 c) {
#ifdef PCDM_EMU
    struct P { u16 a[8], b[8]; } p;
    for (int e = 0; e < 8; ++e) { p.a[e] = a[e]; p.b[e] = b[e]; }
    const char* all = emu::wave_exchange(&p, sizeof(p));
    const int l = emu::lane_id(), j = l & 31, hh = l >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
        float s = c[r];
        for (int k = 0; k < 16; ++k) {
            const P* pa = (const P*)(all + (i + 32 * (k >> 3)) * emu::kSlot);
            const P* pb = (const P*)(all + (j + 32 * (k >> 3)) * emu::kSlot);
            s += bf2f(pa->a[k & 7]) * bf2f(pb->b[k & 7]);
        }
        d[r] = s;
    }
    return d;
#else
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// 16x16x32 bf16:  A lane l -> row (l&15), k = 8*(l>>4)+e ;  B lane l -> col (l&15), same k ;
//                 D lane l, reg r -> col (l&15), row r + 4*(l>>4).
__device__ __forceinline__ f32x4 mfma_16x16x32(u16x8 a, u16x8 b, f32x4 c) {
#ifdef PCDM_EMU
    struct P { u16 a[8], b[8]; } p;
    for (int e = 0; e < 8; ++e) { p.a[e] = a[e]; p.b[e] = b[e]; }
    const char* all = emu::wave_exchange(&p, sizeof(p));
    const int l = emu::lane_id(), j = l & 15, hh = l >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = r + 4 * hh;
        float s = c[r];
        for (int k = 0; k < 32; ++k) {
            const P* pa = (const P*)(all + (i + 16 * (k >> 3)) * emu::kSlot);
            const P* pb = (const P*)(all + (j + 16 * (k >> 3)) * emu::kSlot);
            s += bf2f(pa->a[k & 7]) * bf2f(pb->b[k & 7]);
        }
        d[r] = s;
    }
    return d;
#else
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// ---- OCP fp8 e4m3 (gfx950: e4m3fn -- bias 7, max 448, no infinities; NOT MI300's fnuz) --------------------------------------
// decode / encode on the host side of the emulator and in tests; the GPU converts with v_cvt_pk_fp8_f32 (RNE)
__device__ __forceinline__ float fp8_e4m3_to_f(uint32_t v) {
    const uint32_t s = (v >> 7) & 1, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = (float)m * 0.001953125f;                         // subnormal: m * 2^-9
    else if (e == 15 && m == 7) f = __builtin_nanf("");
    else f = (1.0f + (float)m * 0.125f) * __builtin_bit_cast(float, (uint32_t)((e + 120) << 23));   // 2^(e-7)
    return s ? -f : f;
}
__device__ __forceinline__ uint32_t f_to_fp8_e4m3(float f) {      // RNE, saturating at +-448 (software: emulator / references)
    const uint32_t s = f < 0.f ? 0x80u : 0u;
    float a = f < 0.f ? -f : f;
    if (!(a == a)) return s | 0x7f;
    if (a >= 448.f) return s | 0x7e;
    if (a < 0.0009765625f) return s;                                 // < 2^-10: rounds to zero (ties-to-even at exactly 2^-10 -> 0)
    int e = (int)((__builtin_bit_cast(uint32_t, a) >> 23) & 255) - 127;
    if (e < -6) e = -6;                                              // subnormal range: quantum 2^-9
    const float q = __builtin_bit_cast(float, (uint32_t)((e - 3 + 127) << 23));   // 2^(e-3): one mantissa step
    const float r = __builtin_rintf(a / q);                          // RNE on the grid
    const float v = r * q;
    if (v >= 448.f) return s | 0x7e;
    const uint32_t bits = __builtin_bit_cast(uint32_t, v);
    const int ev = (int)((bits >> 23) & 255) - 127;
    if (v < 0.015625f) return s | (uint32_t)(v * 512.f);             // subnormal: m = v / 2^-9
    return s | (uint32_t)((ev + 7) << 3) | ((bits >> 20) & 7);
}
// four fp32 -> one dword of four e4m3 (byte 0 = first)
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
#ifdef PCDM_EMU
    return f_to_fp8_e4m3(a) | (f_to_fp8_e4m3(b) << 8) | (f_to_fp8_e4m3(c) << 16) | (f_to_fp8_e4m3(d) << 24);
#else
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
#endif
}
// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands e4m3 and unit block scales (E8M0 127 = 2^0): D[i][j] += sum_k A[i][k] B[j][k],
// k = 0..63, at twice the bf16 rate.  A lane l -> row (l&31), its 32 bytes = k 32*(l>>5) .. +31 in order; B likewise with column
// (l&31); D as every 32x32 MFMA: lane l, reg r -> col (l&31), row (r&3) + 8*(r>>2) + 4*(l>>5).
__device__ __forceinline__ f32x16 mfma_f8_32x32x64(u32x8 a, u32x8 b, f32x16 c) {
#ifdef PCDM_EMU
    struct P { uint32_t a[8], b[8]; } p;
    for (int e = 0; e < 8; ++e) { p.a[e] = a[e]; p.b[e] = b[e]; }
    static_assert(sizeof(P) == emu::kSlot, "payload fills one exchange slot");
    const char* all = emu::wave_exchange(&p, sizeof(p));
    const int l = emu::lane_id(), j = l & 31, hh = l >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
        float s = c[r];
        for (int k = 0; k < 64; ++k) {
            const P* pa = (const P*)(all + (i + 32 * (k >> 5)) * emu::kSlot);
            const P* pb = (const P*)(all + (j + 32 * (k >> 5)) * emu::kSlot);
            const int kb = k & 31;
            s += fp8_e4m3_to_f((pa->a[kb >> 2] >> (8 * (kb & 3))) & 255) * fp8_e4m3_to_f((pb->b[kb >> 2] >> (8 * (kb & 3))) & 255);
        }
        d[r] = s;
    }
    return d;
#else
    typedef int i32x8_ __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_bit_cast(i32x8_, a), __builtin_bit_cast(i32x8_, b), c, 0, 0, 0,
                                                           0x7f7f7f7f, 0, 0x7f7f7f7f);
#endif
}

// ---- inter-workgroup hand-off inside one launch (cdna_hip_programming.md Guideline 16) ----------------
#ifdef PCDM_EMU
__device__ __forceinline__ void pcdm_drain_vmem() {}
__device__ __forceinline__ void pcdm_release_agent() {}
__device__ __forceinline__ void pcdm_acquire_agent() {}
__device__ __forceinline__ unsigned pcdm_atomic_inc_agent(unsigned* p) { return (*p)++; }
__device__ __forceinline__ float pcdm_load_agent(const float* p) { return *p; }
__device__ __forceinline__ void pcdm_store_agent(float* p, float v) { *p = v; }
__device__ __forceinline__ unsigned pcdm_load_agent_u32(const unsigned* p) { return *(volatile const unsigned*)p; }
__device__ __forceinline__ void pcdm_store_agent_u32(unsigned* p, unsigned v) { *p = v; }
__device__ __forceinline__ void pcdm_store_sys(float* p, float v) { *p = v; }
__device__ __forceinline__ float pcdm_load_sys(const float* p) { return *p; }
__device__ __forceinline__ unsigned pcdm_load_sys_u32(const unsigned* p) { return *(volatile const unsigned*)p; }
__device__ __forceinline__ void pcdm_store_sys_u32(unsigned* p, unsigned v) { *p = v; }
__device__ __forceinline__ void pcdm_sleep() {}
#else
__device__ __forceinline__ void pcdm_drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pcdm_release_agent() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler may drop the wait behind buffer_wbl2 (G16 pitfall 12)
}
__device__ __forceinline__ void pcdm_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ unsigned pcdm_atomic_inc_agent(unsigned* p) {
    return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pcdm_store_agent(float* p, float v) {   // sc1 write-through store
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float pcdm_load_agent(const float* p) {   // L2-served (bypasses this CU's L1)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned pcdm_load_agent_u32(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pcdm_store_agent_u32(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// SYSTEM scope (sc0 sc1) on both sides of a cross-workgroup hand-off without fences: the store is written through, the load is
// served by memory, not by this XCD's L2.  An agent-scope (sc1) load bypasses only the CU's L1, and the per-XCD L2s are not coherent
// with each other: a line this XCD's L2 still holds from an earlier read of the same address may be returned after another XCD has
// written it (cdna_hip_programming.md G16 lists "sc0 sc1 stores AND loads, both sides" as the fence-free form that is valid).
__device__ __forceinline__ void pcdm_store_sys(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float pcdm_load_sys(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned pcdm_load_sys_u32(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void pcdm_store_sys_u32(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void pcdm_sleep() { __builtin_amdgcn_s_sleep(2); }
#endif

// ---- wave reductions -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#ifdef PCDM_EMU
    return emu_wave_reduce(v, false);   // one exchange instead of six (summation order differs from the butterfly in the last bits)
#else
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef PCDM_EMU
    return emu_wave_reduce(v, true);
#else
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
#endif
}

// any lane's predicate true? (wave-uniform result: s_cmp on the ballot mask)
__device__ __forceinline__ bool wave_any(bool v) {
#ifdef PCDM_EMU
    return wave_max(v ? 1.f : 0.f) > 0.f;
#else
    return __builtin_amdgcn_ballot_w64(v) != 0;
#endif
}

#define PCDM_CHECK_LAUNCH()                          \
    do {                                             \
        hipError_t e_ = hipGetLastError();           \
        if (e_ != hipSuccess) return -(int)e_ - 1000; \
    } while (0)
