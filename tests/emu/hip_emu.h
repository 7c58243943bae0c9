// Lane-level emulator of the small HIP/gfx950 subset the pcdm kernels use.  TEST INFRASTRUCTURE.
//
// The dev container has no GPU, so tests/ build pcdms_amd/csrc/*.hip a second time with
//   clang++ -x c++ -DPCDM_EMU -include tests/emu/hip_emu.h
// into a CPU shared object exposing the same C-ABI.  Every GPU thread becomes a ucontext fiber;
// __syncthreads(), wave shuffles and MFMA are rendezvous points with the documented gfx950
// semantics (wave = 64 lanes, MFMA fragment maps of cdna_hip_programming.md §3).  It checks the
// kernels' index logic against the oracle on tiny shapes; it is never loaded by the product
// (pcdms_amd/_lib.py only loads the hipcc-built libpcdm.so) and is not a fallback.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
extern dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
extern char* dyn_smem;
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
void syncthreads();
// Deposit `bytes` of this lane's payload, wait for the whole wave, return pointer to the wave's
// 64 payload slots (slot stride = kSlot bytes).  Valid until this lane's next-but-one exchange.
constexpr int kSlot = 64;
const char* wave_exchange(const void* payload, int bytes);
int lane_id();
}  // namespace emu

#define threadIdx (emu::threadIdx_)
#define blockIdx (emu::blockIdx_)
#define blockDim (emu::blockDim_)
#define gridDim (emu::gridDim_)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
inline void __syncthreads() { emu::syncthreads(); }
inline float __expf(float x) { return expf(x); }
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
template <class F> inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }

template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    static_assert(sizeof(T) <= 8, "shfl payload");
    const char* all = emu::wave_exchange(&v, sizeof(T));
    T r;
    memcpy(&r, all + ((emu::lane_id() ^ mask) & 63) * emu::kSlot, sizeof(T));
    return r;
}
// whole-wave reductions in ONE exchange (the butterfly of __shfl_xor steps costs log2(64) fiber round trips per lane)
inline float emu_wave_reduce(float v, bool is_max) {
    const char* all = emu::wave_exchange(&v, sizeof(float));
    float acc = 0.f;
    for (int l = 0; l < 64; ++l) {
        float r;
        memcpy(&r, all + l * emu::kSlot, sizeof(float));
        acc = l == 0 ? r : (is_max ? (r > acc ? r : acc) : acc + r);
    }
    return acc;
}
template <class T> inline T __shfl(T v, int src, int width = 64) {
    const char* all = emu::wave_exchange(&v, sizeof(T));
    T r;
    memcpy(&r, all + (src & 63) * emu::kSlot, sizeof(T));
    return r;
}
template <class T> inline T __shfl_down(T v, int d, int width = 64) {
    const char* all = emu::wave_exchange(&v, sizeof(T));
    int s = emu::lane_id() + d;
    if (s > 63) s = emu::lane_id();
    T r;
    memcpy(&r, all + s * emu::kSlot, sizeof(T));
    return r;
}
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
