// What v_permlane32_swap / v_permlane16_swap / DPP row_ror do on gfx950, printed lane by lane (the semantics gemm.hip's statistics
// reduction relies on: pcdm_device.h pcdm_swap32 / pcdm_swap16 / pcdm_row_ror).   hipcc --offload-arch=gfx950 tools/probe_permlane.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    const unsigned a = 100 + lane, b = 200 + lane;
    u32x2_ r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + lane] = r[0]; out[192 + lane] = r[1];
    out[256 + lane] = __builtin_amdgcn_update_dpp(0u, a, 0x128, 0xf, 0xf, false);
    out[320 + lane] = __builtin_amdgcn_update_dpp(0u, a, 0x124, 0xf, 0xf, false);
    out[384 + lane] = __builtin_amdgcn_update_dpp(0u, a, 0x122, 0xf, 0xf, false);
}
int main() {
    unsigned* d; unsigned h[448];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[7] = {"swap32 a'", "swap32 b'", "swap16 a'", "swap16 b'", "ror8(a)", "ror4(a)", "ror2(a)"};
    for (int i = 0; i < 7; ++i) {
        printf("%s:", names[i]);
        for (int l = 0; l < 64; ++l) printf(" %u", h[i * 64 + l]);
        printf("\n");
    }
    return 0;
}
