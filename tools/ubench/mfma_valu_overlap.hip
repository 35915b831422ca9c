// Does the FP64 matrix instruction run BESIDE the FP64 vector instructions of another wave of the same SIMD on gfx950?
// (tools, not product.) DESIGN.md section 9 assumed it does -- "the matrix pipe running beside the vector pipe" -- and the
// staggered form of the shared-matrix dense kernel (tools/experiments/coop_stagger.patch) was built on that; it measured
// 1.85e8 against the lock-step form's 2.67e8 leapfrog-steps/s. This probe asks the hardware directly: ONE workgroup of
// eight waves per CU, two of them given a role for a fixed time, the rest idle:
//   m = v_mfma_f64_16x16x4_f64 (four independent accumulators), M = v_mfma_f32_32x32x2_f32 (two accumulators),
//   d = v_fma_f64 (eight independent accumulators), s = v_fma_f32 (eight), - = idle; m2 / m1 = the matrix instruction on two
//   alternating accumulators (the product loop's shape) / one; mp = m2 with 64 idle cycles after every issue; D = d at
//   s_setprio 3; mi / Mi = ONE wave alternating a matrix instruction with eight independent vector ones (clocks per group).
// Output per configuration: SIMD id of the two waves (HW_REG_HW_ID) and shader cycles per instruction of each.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_valu_overlap mfma_valu_overlap.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

enum Role : int { kIdle = 0, kMfma64 = 1, kFma64 = 2, kFma32 = 3, kMfma32 = 4, kMfma64Two = 5, kMfma64Dep = 6, kFma64Hi = 7, kMfma64Paced = 8, kInterleave64 = 9, kInterleave32 = 10 };

__global__ __launch_bounds__(512) void probe(long long* out, int role_a, int wave_b, int role_b, long long budget, double seed) {
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int role = wave == 0 ? role_a : (wave == wave_b ? role_b : kIdle);
    const int simd = __builtin_amdgcn_s_getreg(((2 - 1) << 11) | (4 << 6) | 4);   // HW_REG_HW_ID[5:4]
    __syncthreads();
    long long n = 0;
    const long long t0 = clock64();
    long long t1 = t0;
    if (role == kFma64Hi) __builtin_amdgcn_s_setprio(3);   // the vector wave ahead of the matrix wave in the issue arbiter
    if (role == kMfma64Two || role == kMfma64Dep || role == kMfma64Paced) {   // the product loop's shape: two alternating accumulators / one
        v4d c0 = {0, 0, 0, 0}, c1 = c0;
        const double a = seed + threadIdx.x, b = 1.0 / (1.0 + threadIdx.x);
        while (t1 - t0 < budget) {
            if (role == kMfma64Two) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                }
            } else if (role == kMfma64Dep) {
#pragma unroll
                for (int u = 0; u < 16; ++u) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            } else {   // paced: the wave does not ask for the matrix pipe while it is busy (64 idle cycles after each issue)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 13");
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 13");
                }
            }
            n += 16;
            t1 = clock64();
        }
        if (c0[0] + c1[1] == 12345.678) out[0] = 1;
    } else if (role == kMfma64) {
        v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const double a = seed + threadIdx.x, b = 1.0 / (1.0 + threadIdx.x);
        while (t1 - t0 < budget) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            }
            n += 16;
            t1 = clock64();
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678) out[0] = 1;
    } else if (role == kInterleave64) {   // ONE wave: every matrix instruction followed by eight independent vector ones
        v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const double a = seed + threadIdx.x, b = 1.0 / (1.0 + threadIdx.x);
        double x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
        const double m = 0.9999999, k = 1e-7 * threadIdx.x;
#define LMC_EIGHT_FMA64 asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n" \
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n" \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(k));
        while (t1 - t0 < budget) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); LMC_EIGHT_FMA64
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0); LMC_EIGHT_FMA64
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); LMC_EIGHT_FMA64
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0); LMC_EIGHT_FMA64
            }
            n += 16;
            t1 = clock64();
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345.678) out[0] = 1;
    } else if (role == kInterleave32) {   // the same with the float32 matrix instruction and float32 vector ones
        v16f c0 = {0}, c1 = {0};
        const float a = static_cast<float>(seed) + threadIdx.x, b = 1.0f / (1.0f + threadIdx.x);
        float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
        const float m = 0.99999f, k = 1e-5f * threadIdx.x;
#define LMC_EIGHT_FMA32 asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(k));
        while (t1 - t0 < budget) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0); LMC_EIGHT_FMA32
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0); LMC_EIGHT_FMA32
            }
            n += 16;
            t1 = clock64();
        }
        if (c0[0] + c1[1] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345.678f) out[0] = 1;
    } else if (role == kMfma32) {
        v16f c0 = {0}, c1 = {0};
        const float a = static_cast<float>(seed) + threadIdx.x, b = 1.0f / (1.0f + threadIdx.x);
        while (t1 - t0 < budget) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            }
            n += 16;
            t1 = clock64();
        }
        if (c0[0] + c1[1] == 12345.678f) out[0] = 1;
    } else if (role == kFma64 || role == kFma64Hi) {
        double x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
        const double m = 0.9999999, k = 1e-7 * threadIdx.x;
        while (t1 - t0 < budget) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(k));
            }
            n += 256;
            t1 = clock64();
        }
        if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345.678) out[0] = 1;
    } else if (role == kFma32) {
        float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
        const float m = 0.99999f, k = 1e-5f * threadIdx.x;
        while (t1 - t0 < budget) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(k));
            }
            n += 256;
            t1 = clock64();
        }
        if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345.678f) out[0] = 1;
    }
    if ((threadIdx.x & 63) == 0) {
        long long* o = out + 1 + (static_cast<long long>(blockIdx.x) * 8 + wave) * 3;
        o[0] = n; o[1] = t1 - t0; o[2] = simd;
    }
}

int main() {
    int n_cu = 0;
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    const int blocks = n_cu;   // one 512-thread workgroup per CU (they all run at once: the chip-wide clock / power state is the loaded one)
    long long* d_out;
    const size_t n_out = 1 + static_cast<size_t>(blocks) * 8 * 3;
    CHECK(hipMalloc(&d_out, n_out * sizeof(long long)));
    std::vector<long long> h(n_out);
    struct Cfg { const char* name; int role_a, wave_b, role_b; };
    const Cfg cfgs[] = {
        {"m -      (f64 matrix alone)", kMfma64, 4, kIdle},
        {"d -      (f64 vector alone)", kFma64, 4, kIdle},
        {"s -      (f32 vector alone)", kFma32, 4, kIdle},
        {"M -      (f32 matrix alone)", kMfma32, 4, kIdle},
        {"m + d on wave 4 (same SIMD?)", kMfma64, 4, kFma64},
        {"m + d on wave 1 (other SIMD?)", kMfma64, 1, kFma64},
        {"m + s on wave 4", kMfma64, 4, kFma32},
        {"m + m on wave 4", kMfma64, 4, kMfma64},
        {"d + d on wave 4", kFma64, 4, kFma64},
        {"M + d on wave 4", kMfma32, 4, kFma64},
        {"M + s on wave 4", kMfma32, 4, kFma32},
        {"m2 -     (two accumulators)", kMfma64Two, 4, kIdle},
        {"m1 -     (one accumulator)", kMfma64Dep, 4, kIdle},
        {"mp -     (paced, 2 acc.)", kMfma64Paced, 4, kIdle},
        {"m2 + d on wave 4", kMfma64Two, 4, kFma64},
        {"m1 + d on wave 4", kMfma64Dep, 4, kFma64},
        {"mp + d on wave 4", kMfma64Paced, 4, kFma64},
        {"m + D (s_setprio 3) on wave 4", kMfma64, 4, kFma64Hi},
        {"m2 + D (s_setprio 3) on wave 4", kMfma64Two, 4, kFma64Hi},
        {"d + D (s_setprio 3) on wave 4", kFma64, 4, kFma64Hi},
        {"mi -  one wave: m, 8 d, m, 8 d ..", kInterleave64, 4, kIdle},
        {"Mi -  one wave: M, 8 s, M, 8 s ..", kInterleave32, 4, kIdle},
    };
    const char* rname[] = {"-", "v_mfma_f64_16x16x4_f64", "v_fma_f64", "v_fma_f32", "v_mfma_f32_32x32x2_f32", "mfma_f64 2 acc.", "mfma_f64 1 acc.", "v_fma_f64 prio 3", "mfma_f64 paced", "mfma_f64 + 8 v_fma_f64", "mfma_f32 + 8 v_fma_f32"};
    for (const Cfg& c : cfgs) {
        CHECK(hipMemset(d_out, 0, n_out * sizeof(long long)));
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, d_out, c.role_a, c.wave_b, c.role_b, 4000000LL, 1.5);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d_out, n_out * sizeof(long long), hipMemcpyDeviceToHost));
        double cyc_a = 0, cyc_b = 0;
        int same = 0;
        for (int b = 0; b < blocks; ++b) {
            const long long* a = &h[1 + (static_cast<size_t>(b) * 8 + 0) * 3];
            const long long* w = &h[1 + (static_cast<size_t>(b) * 8 + c.wave_b) * 3];
            cyc_a += a[0] ? static_cast<double>(a[1]) / a[0] : 0.0;
            cyc_b += w[0] ? static_cast<double>(w[1]) / w[0] : 0.0;
            same += a[2] == w[2];
        }
        printf("%-32s wave 0: %-24s %7.2f clocks/instr | wave %d: %-24s %7.2f clocks/instr | same SIMD in %d of %d workgroups\n", c.name,
               rname[c.role_a], cyc_a / blocks, c.wave_b, rname[c.role_b], cyc_b / blocks, same, blocks);
    }
    CHECK(hipFree(d_out));
    return 0;
}
