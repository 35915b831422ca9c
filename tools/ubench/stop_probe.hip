// How can the host reach a GPU whose every wave slot is held by persistent kernels? (lmc_engine_request_stop)
// Fills all wave slots with waves that poll a device word once per "iteration" (system-scope relaxed atomic load), then
// sets the word by different means and measures how long the kernels take to notice.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stop_probe.hip -o tools/ubench/stop_probe && tools/ubench/stop_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64, 4) void spin(const int* stop, long long* iters, long long max_iters) {
    long long n = 0;
    double x = threadIdx.x;
    for (; n < max_iters; ++n) {
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0) break;
        for (int k = 0; k < 2000; ++k) x = x * 1.0000001 + 1e-9;   // ~ one short NUTS iteration of arithmetic
    }
    if (threadIdx.x == 0) iters[blockIdx.x] = n + (x < 0 ? 1 : 0);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    int* flag; long long* iters; int* pinned;
    const int grid = 4096 * 4;    // 4 rounds of full residency on each of two streams
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&iters, grid * 8));
    CK(hipHostMalloc(&pinned, 4)); *pinned = 1;
    hipStream_t s[2], ctl;
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    int lo_p = 0, hi_p = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    hipStream_t ctl_hi;
    CK(hipStreamCreateWithFlags(&ctl, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&ctl_hi, hipStreamNonBlocking, hi_p));
    printf("stream priorities: lowest %d, highest %d\n", lo_p, hi_p);
    static const int one = 1;
    hipStream_t fan[4];
    for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithFlags(&fan[i], hipStreamNonBlocking));
    for (int method = 0; method < 13; ++method) {
        CK(hipMemset(flag, 0, 4)); CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int r = 0; r < 3; ++r)
            for (int b = 0; b < 2; ++b) hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s[b], flag, iters, 20000LL);
        std::this_thread::sleep_for(std::chrono::milliseconds(30));
        const double t1 = now();
        const char* name = "";
        if (method == 0) { name = "hipMemcpyAsync pageable"; CK(hipMemcpyAsync(flag, &one, 4, hipMemcpyHostToDevice, ctl)); }
        if (method == 1) { name = "hipMemcpyAsync pinned"; CK(hipMemcpyAsync(flag, pinned, 4, hipMemcpyHostToDevice, ctl)); }
        if (method == 2) { name = "hipStreamWriteValue32"; CK(hipStreamWriteValue32(ctl, flag, 1, 0)); }
        if (method == 3) { name = "hipMemsetAsync"; CK(hipMemsetAsync(flag, 1, 4, ctl)); }
        if (method >= 4 && method < 7) { name = "WriteValue32, high-priority stream"; CK(hipStreamWriteValue32(ctl_hi, flag, 1, 0)); }
        if (method >= 7 && method < 9) { name = "WriteValue32 again"; CK(hipStreamWriteValue32(ctl, flag, 1, 0)); }
        if (method >= 9) {
            name = "WriteValue32 on 4 streams, first to land";
            for (int i = 0; i < 4; ++i) CK(hipStreamWriteValue32(fan[i], flag, 1, 0));
            bool landed = false;
            while (!landed) for (int i = 0; i < 4 && !landed; ++i) landed = hipStreamQuery(fan[i]) == hipSuccess;
        } else
        CK(hipStreamSynchronize(method >= 4 && method < 7 ? ctl_hi : ctl));
        const double t2 = now();
        CK(hipDeviceSynchronize());
        const double t3 = now();
        printf("%-26s request returned after %.3f ms, kernels drained %.3f ms after the request (job ran %.1f ms before it)\n", name,
               1e3 * (t2 - t1), 1e3 * (t3 - t1), 1e3 * (t1 - t0));
    }
    // without any request: how long the job takes
    CK(hipMemset(flag, 0, 4)); CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int b = 0; b < 2; ++b) hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s[b], flag, iters, 2000LL);
    CK(hipDeviceSynchronize());
    printf("2000 iterations of 2 x %d waves, no request: %.1f ms\n", grid, 1e3 * (now() - t0));
    return 0;
}
