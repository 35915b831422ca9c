// Issue-cost microbenchmark for the instruction mix of lmc::run_kernel on gfx950 (tools, not product).
// Each kernel runs ITER x 16 copies of one instruction pattern per wave; W blocks of 256 threads per CU
// give W waves per SIMD. Output: shader cycles per instruction per SIMD (elapsed / (ITER*16*W)).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_cost valu_cost.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP16(S) S S S S S S S S S S S S S S S S
#define REP8(S) S S S S S S S S
#define REP4(S) S S S S

#define KERNEL(NAME, BODY16, NINST) \
__global__ __launch_bounds__(256) void NAME(long long* out, int iters, double seed) { \
    __shared__ double lds[4096]; \
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    double b0 = 1.0000001, b1 = 0.9999999; \
    unsigned addr = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 4096; \
    unsigned addr16 = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096; \
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1; __syncthreads(); \
    int sidx = __builtin_amdgcn_readfirstlane(iters & 31); \
    asm volatile("v_mov_b32 v220, %0\n v_mov_b32 v221, %1\n s_mov_b32 s44, %2\n" \
                 "v_cvt_f64_u32 v[200:201], %0\n v_cvt_f64_u32 v[202:203], %1\n v_cvt_f64_u32 v[204:205], %0\n v_cvt_f64_u32 v[206:207], %1\n" \
                 "v_cvt_f64_u32 v[208:209], %0\n v_cvt_f64_u32 v[210:211], %1\n v_cvt_f64_u32 v[212:213], %0\n v_cvt_f64_u32 v[214:215], %1\n" \
                 "v_mov_b32 v216, 0\n v_mov_b32 v217, 0x3ff00000\n v_mov_b32 v218, 0\n v_mov_b32 v219, 0x3ff00000\n" \
                 :: "v"(addr), "v"(addr16), "s"(sidx) : "v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","s44"); \
    long long t0 = clock64(); \
    for (int i = 0; i < iters; ++i) { \
        asm volatile(BODY16 ::: "memory", "s40", "s41", "s42", "s43", "vcc", \
            "v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215", \
            "v216","v217","v218","v219"); \
    } \
    long long t1 = clock64(); \
    double s; asm volatile("v_mov_b64 %0, v[200:201]" : "=v"(s) :: "v200","v201"); \
    if (s == 12345.678) out[0] = 1; \
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; \
}

// operands: v[200:201]..v[214:215] = a0..a7 (64-bit VGPR pairs), v[216:217] b0, v[218:219] b1, v220 addr, v221 addr16, s44 sidx
// --- independent f64 ops (8 accumulators round-robin)
KERNEL(k_add_f64, REP4("v_add_f64 v[200:201], v[200:201], v[216:217]\n v_add_f64 v[202:203], v[202:203], v[216:217]\n v_add_f64 v[204:205], v[204:205], v[216:217]\n v_add_f64 v[206:207], v[206:207], v[216:217]\n"), 16)
KERNEL(k_mul_f64, REP4("v_mul_f64 v[200:201], v[200:201], v[216:217]\n v_mul_f64 v[202:203], v[202:203], v[218:219]\n v_mul_f64 v[204:205], v[204:205], v[216:217]\n v_mul_f64 v[206:207], v[206:207], v[218:219]\n"), 16)
KERNEL(k_fma_f64, REP4("v_fma_f64 v[200:201], v[200:201], v[216:217], v[218:219]\n v_fma_f64 v[202:203], v[202:203], v[218:219], v[216:217]\n v_fma_f64 v[204:205], v[204:205], v[216:217], v[218:219]\n v_fma_f64 v[206:207], v[206:207], v[218:219], v[216:217]\n"), 16)
KERNEL(k_add_f64_dep, REP16("v_add_f64 v[200:201], v[200:201], v[216:217]\n"), 16)
// --- 32-bit moves
KERNEL(k_mov_b32, REP4("v_mov_b32 v200, v202\n v_mov_b32 v204, v206\n v_mov_b32 v208, v210\n v_mov_b32 v212, v214\n"), 16)
KERNEL(k_mov_b64, REP4("v_mov_b64 v[200:201], v[202:203]\n v_mov_b64 v[204:205], v[206:207]\n v_mov_b64 v[208:209], v[210:211]\n v_mov_b64 v[212:213], v[214:215]\n"), 16)
KERNEL(k_mov_dpp, REP4("v_mov_b32_dpp v200, v202 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v204, v206 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v208, v210 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v212, v214 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"), 16)
KERNEL(k_add_f32, REP4("v_add_f32 v200, v200, v216\n v_add_f32 v202, v202, v216\n v_add_f32 v204, v204, v216\n v_add_f32 v206, v206, v216\n"), 16)
KERNEL(k_add_f32_dpp, REP4("v_add_f32_dpp v200, v200, v200 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp v202, v202, v202 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp v204, v204, v204 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp v206, v206, v206 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"), 16)
KERNEL(k_pk_add_f32, REP4("v_pk_add_f32 v[200:201], v[200:201], v[216:217]\n v_pk_add_f32 v[202:203], v[202:203], v[216:217]\n v_pk_add_f32 v[204:205], v[204:205], v[216:217]\n v_pk_add_f32 v[206:207], v[206:207], v[216:217]\n"), 16)
KERNEL(k_cndmask, REP4("v_cndmask_b32 v200, v202, v204, vcc\n v_cndmask_b32 v206, v208, v210, vcc\n v_cndmask_b32 v212, v214, v202, vcc\n v_cndmask_b32 v204, v206, v208, vcc\n"), 16)
KERNEL(k_permlane32_swap, REP4("v_permlane32_swap_b32 v200, v202\n s_nop 1\n v_permlane32_swap_b32 v204, v206\n s_nop 1\n v_permlane32_swap_b32 v208, v210\n s_nop 1\n v_permlane32_swap_b32 v212, v214\n s_nop 1\n"), 16)
KERNEL(k_readlane, REP4("v_readlane_b32 s40, v200, 5\n v_readlane_b32 s41, v202, 6\n v_readlane_b32 s42, v204, 7\n v_readlane_b32 s43, v206, 8\n"), 16)
KERNEL(k_readlane_s, REP4("v_readlane_b32 s40, v200, s44\n v_readlane_b32 s41, v202, s44\n v_readlane_b32 s42, v204, s44\n v_readlane_b32 s43, v206, s44\n"), 16)
KERNEL(k_readfirstlane, REP4("v_readfirstlane_b32 s40, v200\n v_readfirstlane_b32 s41, v202\n v_readfirstlane_b32 s42, v204\n v_readfirstlane_b32 s43, v206\n"), 16)
KERNEL(k_cmp_f64, REP4("v_cmp_lt_f64 vcc, v[200:201], v[202:203]\n v_cmp_lt_f64 vcc, v[204:205], v[206:207]\n v_cmp_lt_f64 vcc, v[208:209], v[210:211]\n v_cmp_lt_f64 vcc, v[212:213], v[214:215]\n"), 16)
KERNEL(k_rcp_f64, REP4("v_rcp_f64 v[200:201], v[200:201]\n v_rcp_f64 v[202:203], v[202:203]\n v_rcp_f64 v[204:205], v[204:205]\n v_rcp_f64 v[206:207], v[206:207]\n"), 16)
KERNEL(k_exp_f32, REP4("v_exp_f32 v200, v200\n v_exp_f32 v202, v202\n v_exp_f32 v204, v204\n v_exp_f32 v206, v206\n"), 16)
KERNEL(k_ldexp_f64, REP4("v_ldexp_f64 v[200:201], v[200:201], v216\n v_ldexp_f64 v[202:203], v[202:203], v216\n v_ldexp_f64 v[204:205], v[204:205], v216\n v_ldexp_f64 v[206:207], v[206:207], v216\n"), 16)
KERNEL(k_salu, REP4("s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n"), 16)
// --- reduction stage as compiled today: 2 DPP movs + 1 f64 add, dependent chain (16 "instructions" = 5.33 stages)
KERNEL(k_stage_dpp, REP4("v_mov_b32_dpp v202, v200 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v203, v201 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f64 v[200:201], v[200:201], v[202:203]\n s_nop 1\n"), 12)
// two independent chains interleaved
KERNEL(k_stage_dpp2, REP4("v_mov_b32_dpp v202, v200 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v203, v201 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v206, v204 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v207, v205 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f64 v[200:201], v[200:201], v[202:203]\n v_add_f64 v[204:205], v[204:205], v[206:207]\n"), 24)
// --- LDS-pipe cross-lane: ds_swizzle (xor butterfly), ds_bpermute
KERNEL(k_ds_swizzle, REP4("ds_swizzle_b32 v200, v200 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 v202, v202 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 v204, v204 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 v206, v206 offset:swizzle(BITMASK_PERM,\"0000p\")\n") "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_ds_bpermute, REP4("ds_bpermute_b32 v200, v220, v200\n ds_bpermute_b32 v202, v220, v202\n ds_bpermute_b32 v204, v220, v204\n ds_bpermute_b32 v206, v220, v206\n") "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_stage_swz, REP4("ds_swizzle_b32 v202, v200 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 v203, v201 offset:swizzle(BITMASK_PERM,\"0000p\")\n s_waitcnt lgkmcnt(0)\n v_add_f64 v[200:201], v[200:201], v[202:203]\n"), 12)
// --- LDS memory
KERNEL(k_ds_read_b64, REP4("ds_read_b64 v[200:201], v220\n ds_read_b64 v[202:203], v220 offset:512\n ds_read_b64 v[204:205], v220 offset:1024\n ds_read_b64 v[206:207], v220 offset:1536\n") "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_ds_read_b128, REP4("ds_read_b128 v[200:203], v221\n ds_read_b128 v[204:207], v221 offset:1024\n ds_read_b128 v[208:211], v221 offset:2048\n ds_read_b128 v[212:215], v221 offset:3072\n") "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_ds_write_b64, REP4("ds_write_b64 v220, v[200:201]\n ds_write_b64 v220, v[202:203] offset:512\n ds_write_b64 v220, v[204:205] offset:1024\n ds_write_b64 v220, v[206:207] offset:1536\n") "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_ds_write_b128, REP4("ds_write_b128 v221, v[200:203]\n ds_write_b128 v221, v[204:207] offset:1024\n ds_write_b128 v221, v[208:211] offset:2048\n ds_write_b128 v221, v[212:215] offset:3072\n") "s_waitcnt lgkmcnt(0)\n", 16)
// --- mixes: does LDS traffic from the same wave overlap its own VALU stream?
KERNEL(k_mix_add_dsread, REP4("ds_read_b64 v[208:209], v220\n v_add_f64 v[200:201], v[200:201], v[216:217]\n v_add_f64 v[202:203], v[202:203], v[216:217]\n v_add_f64 v[204:205], v[204:205], v[216:217]\n") "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_mix_add_dswrite, REP4("ds_write_b64 v220, v[208:209]\n v_add_f64 v[200:201], v[200:201], v[216:217]\n v_add_f64 v[202:203], v[202:203], v[216:217]\n v_add_f64 v[204:205], v[204:205], v[216:217]\n") "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_mix_add_salu, REP4("s_add_u32 s40, s40, 1\n v_add_f64 v[200:201], v[200:201], v[216:217]\n s_add_u32 s41, s41, 1\n v_add_f64 v[202:203], v[202:203], v[216:217]\n"), 16)
KERNEL(k_mix_add_mov, REP4("v_mov_b32 v208, v210\n v_add_f64 v[200:201], v[200:201], v[216:217]\n v_mov_b32 v212, v214\n v_add_f64 v[202:203], v[202:203], v[216:217]\n"), 16)

struct Entry { const char* name; void (*fn)(long long*, int, double); int ninst; };
#define E(NAME, N) {#NAME, NAME, N}

int main(int argc, char** argv) {
    int iters = 2000;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    long long* out; CHECK(hipMalloc(&out, sizeof(long long) * (1 + 4 * cus * 8)));
    std::vector<Entry> es = {
        E(k_add_f64,16), E(k_mul_f64,16), E(k_fma_f64,16), E(k_add_f64_dep,16), E(k_mov_b32,16), E(k_mov_b64,16), E(k_mov_dpp,16),
        E(k_add_f32,16), E(k_add_f32_dpp,16), E(k_pk_add_f32,16), E(k_cndmask,16), E(k_permlane32_swap,16), E(k_readlane,16), E(k_readlane_s,16),
        E(k_readfirstlane,16), E(k_cmp_f64,16), E(k_rcp_f64,16), E(k_exp_f32,16), E(k_ldexp_f64,16), E(k_salu,16),
        E(k_stage_dpp,12), E(k_stage_dpp2,24), E(k_ds_swizzle,16), E(k_ds_bpermute,16), E(k_stage_swz,12),
        E(k_ds_read_b64,16), E(k_ds_read_b128,16), E(k_ds_write_b64,16), E(k_ds_write_b128,16),
        E(k_mix_add_dsread,16), E(k_mix_add_dswrite,16), E(k_mix_add_salu,16), E(k_mix_add_mov,16)};
    printf("%-22s %8s %8s %8s %8s   (shader cycles per instruction per SIMD at W waves/SIMD; per-wave latency in parentheses at W=1)\n", "pattern", "W=1", "W=2", "W=3", "W=4");
    for (auto& e : es) {
        printf("%-22s", e.name);
        for (int W = 1; W <= 4; ++W) {
            const int blocks = cus * W;
            CHECK(hipMemset(out, 0, sizeof(long long) * (1 + 4 * blocks)));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0);   // warm
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
            CHECK(hipDeviceSynchronize());
            std::vector<long long> h(1 + 4 * blocks);
            CHECK(hipMemcpy(h.data(), out, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
            std::vector<long long> v(h.begin() + 1, h.end());
            std::sort(v.begin(), v.end());
            const double med = (double)v[v.size() / 2];
            printf(" %8.2f", med / ((double)iters * e.ninst * W));
        }
        printf("\n");
    }
    return 0;
}
