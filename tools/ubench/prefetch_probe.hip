// Feasibility probe (round 6): a low-footprint "weight prefetch" kernel -- one wave per CU streaming a tensor through the memory hierarchy so that
// it sits in the Infinity Cache / L2 when its consumer starts -- run CONCURRENTLY with a dense GEMM on another stream: what does the GEMM lose?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/prefetch_probe.hip -o tools/ubench/libprefetch_probe.so   (tools/probe_prefetch.py)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void prefetch_kernel(const u32x4* __restrict__ p, int64_t n16, uint32_t* sink) {
    const int64_t stride = (int64_t)gridDim.x * 64;
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x; i < n16; i += stride * 8) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i + k * stride < n16) ? p[i + k * stride] : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;   // (never in practice: keeps the loads)
}

extern "C" int prefetch_probe(const void* p, int64_t bytes, int waves, void* sink, void* stream) {
    prefetch_kernel<<<dim3(waves), dim3(64), 0, (hipStream_t)stream>>>((const u32x4*)p, bytes / 16, (uint32_t*)sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
