// Prints what v_permlane16_swap / v_permlane32_swap / DPP row_shl do to lane ids (gfx950 probe; tools, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned a = lane, b = 100 + lane;
    auto s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = s16[0]; out[64 + lane] = s16[1]; out[128 + lane] = s32[0]; out[192 + lane] = s32[1];
    out[256 + lane] = __builtin_amdgcn_mov_dpp((int)lane, 0x108, 0xf, 0xf, true);
    out[320 + lane] = __builtin_amdgcn_mov_dpp((int)lane, 0x104, 0xf, 0xf, true);
    out[384 + lane] = __builtin_amdgcn_mov_dpp((int)lane, 0x118, 0xf, 0xf, true);
}
int main() {
    unsigned* d; hipMalloc(&d, 448 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned h[448]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"swap16[0]", "swap16[1]", "swap32[0]", "swap32[1]", "row_shl:8", "row_shl:4", "row_shr:8"};
    for (int k = 0; k < 7; ++k) { printf("%-10s", names[k]); for (int l = 0; l < 64; ++l) printf(" %3u", h[k * 64 + l]); printf("\n"); }
    return 0;
}
