// liblmc_hip.so: host side of the C ABI declared in include/lmc_hip.h.
// Owns the per-chain state in HBM, dispatches the (target family x vector width) kernel
// instantiation, and moves results in and out. No torch, no exceptions across the boundary.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/lmc_hip.h"
#include "lmc_sampler.hpp"
#include "lmc_unit_kernels.hpp"
#include "lmc_dense_launch.hpp"
#include "lmc_tick_launch.hpp"
#include "lmc_wide_launch.hpp"
#ifdef LMC_USER_TARGET_HEADER
#include LMC_USER_TARGET_HEADER
#endif

using namespace lmc;

// hipMemcpy(..., hipMemcpyDefault) on pageable host pointers can leave a stale "last error" behind (pointer-attribute
// probing); clear it before a launch so that the hipGetLastError() after the launch reports THIS launch only.
#define LMC_LAUNCH(...)            \
    do {                           \
        (void)hipGetLastError();   \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// unit kernels (share the device functions with run_kernel)
// ---------------------------------------------------------------------------------------------
namespace lmc {

__global__ __launch_bounds__(64) void seed_kernel(ChainArrays A, const uint32_t* seeds) {
    const int c = blockIdx.x;
    RngState r;
    r.mt = A.mt + static_cast<long long>(c) * kMtN;
    mt_seed(r, seeds[c]);
    if (lane_id() == 0) {
        A.rng_pos[c] = r.pos;
        A.rng_has_gauss[c] = 0;
        A.rng_gauss[c] = 0.0;
    }
}

__global__ __launch_bounds__(64) void rng_draw_kernel(ChainArrays A, const int* ops, int n_ops, double* out,
                                                      long long out_stride, double* stage, long long stage_stride) {
    const int c = blockIdx.x;
    const int lane = lane_id();
    RngState r;
    r.mt = A.mt + static_cast<long long>(c) * kMtN;
    r.pos = first_i32(A.rng_pos[c]);
    r.has_gauss = first_i32(A.rng_has_gauss[c]);
    r.gauss = first_f64(A.rng_gauss[c]);
    double* o = out + static_cast<long long>(c) * out_stride;
    for (int k = 0; k < n_ops; ++k) {
        const int op = ops[k];
        if (op > 0) {
            rng_normals(r, op, o, stage + static_cast<long long>(c) * stage_stride);
            o += op;
        } else {
            for (int i = 0; i < -op; ++i) {
                const double u = rng_uniform(r);
                if (lane == 0) o[i] = u;
            }
            o += -op;
        }
    }
    if (lane == 0) {
        A.rng_pos[c] = r.pos;
        A.rng_has_gauss[c] = r.has_gauss;
        A.rng_gauss[c] = r.gauss;
    }
}

template <int NS>
__global__ __launch_bounds__(64) void momentum_kernel(ChainArrays A, int momentum_f32, double* out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = blockIdx.x;
    const int lane = lane_id();
    const int d = A.d;
    RngState r;
    r.mt = A.mt + static_cast<long long>(c) * kMtN;
    r.pos = first_i32(A.rng_pos[c]);
    r.has_gauss = first_i32(A.rng_has_gauss[c]);
    r.gauss = first_f64(A.rng_gauss[c]);
    rng_normals(r, d, lds, lds + A.dpad);
    const long long row = static_cast<long long>(c) * A.dpad;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        if (e < d) {
            const float is = A.inv_std[row + e];
            const double z = lds[e];
            out[static_cast<long long>(c) * d + e] =
                momentum_f32 ? static_cast<double>(is * static_cast<float>(z)) : z * static_cast<double>(is);
        }
    }
    if (lane == 0) {
        A.rng_pos[c] = r.pos;
        A.rng_has_gauss[c] = r.has_gauss;
        A.rng_gauss[c] = r.gauss;
    }
}

// QuadPotentialDiagAdapt.reset() (quadpotential.py:195-204) / QuadPotentialDiag.__init__ (:349-365)
// + DualAverageAdaptation.reset() (step_sizes.py:49-56) + iter_count = 0.
__global__ void reset_kernel(ChainArrays A, const double* init_mean, const float* init_diag, const double* init_diag64,
                             int mass_f64, double init_weight,
                             int adapt, double log_step0, double mu, int reset_step, int reset_mass, int window) {
    const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long n = static_cast<long long>(A.chains) * A.dpad;
    if (idx >= n) return;
    const int c = static_cast<int>(idx / A.dpad);
    const int e = static_cast<int>(idx % A.dpad);
    if (reset_mass) {
        const float diag = (e < A.d) ? init_diag[idx] : 1.0f;
        const float sd = sqrtf(diag);
        A.var[idx] = diag;
        A.inv_std[idx] = 1.0f / sd;
        double diag_d = static_cast<double>(diag);
        if (A.var64 != nullptr) {   // wide kernels: the diagonal in float64 (float32-valued unless the potential's dtype is float64)
            diag_d = (e < A.d) ? init_diag64[idx] : 1.0;
            A.var64[idx] = diag_d;
            A.inv_std64[idx] = mass_f64 ? 1.0 / sqrt(diag_d) : static_cast<double>(1.0f / sd);
        }
        if (adapt) {
            // foreground: mean = initial_mean, raw_var = initial_diag * weight; background: zeros
            A.wmean[idx] = (e < A.d) ? init_mean[idx] : 0.0;
            A.wraw[idx] = (e < A.d) ? diag_d * init_weight : 0.0;
            A.wmean[n + idx] = 0.0;
            A.wraw[n + idx] = 0.0;
        }
        if (e == 0) {
            A.wsum[c * 2 + 0] = init_weight;
            A.wsum[c * 2 + 1] = 0.0;
            A.wsel[c] = 0;
            A.n_samples[c] = 0;
            A.awindow[c] = window;
        }
    }
    if (reset_step && A.mom_mean != nullptr) {
        A.mom_mean[idx] = 0.0;
        A.mom_m2[idx] = 0.0;
        if (e == 0) A.mom_n[c] = 0;
    }
    if (reset_step && e == 0) {
        A.status[c] = 0;   // a chain stopped by "Bad initial energy" starts afresh (the reference raises per call and recovers)
        A.da[c * 4 + 0] = log_step0;
        A.da[c * 4 + 1] = log_step0;
        A.da[c * 4 + 2] = 0.0;
        A.da[c * 4 + 3] = mu;
        A.da_count[c] = 1;
        A.iter_count[c] = 0;
    }
}

__global__ void set_da_kernel(ChainArrays A, double log_step, double log_bar, double hbar, int count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.chains) return;
    A.da[c * 4 + 0] = log_step;
    A.da[c * 4 + 1] = log_bar;
    A.da[c * 4 + 2] = hbar;
    A.da_count[c] = count;
}

// QuadPotentialDiagAdapt.update(sample = the chain's current position, grad, tune = True) as a call of its own
// (quadpotential.py:231-245): the very device function the sampling kernel runs after every tuning iteration.
template <int NS>
__global__ __launch_bounds__(64) void mass_update_kernel(ChainArrays A, SamplerParams P) {
    const int c = blockIdx.x;
    const int tid = lane_id();
    const long long row = static_cast<long long>(c) * A.dpad;
    double q[NS], vard[NS];
    float var[NS], inv_std[NS];
    vload<NS>(A.q + row, q);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        var[s] = A.var[row + tid * NS + s];
        inv_std[s] = A.inv_std[row + tid * NS + s];
        vard[s] = static_cast<double>(var[s]);
    }
    MassScalars ms;
    ms.n_samples = first_i32(A.n_samples[c]);
    ms.wsel = first_i32(A.wsel[c]);
    ms.wsum_f = first_f64(A.wsum[c * 2 + ms.wsel]);
    ms.wsum_b = first_f64(A.wsum[c * 2 + (1 - ms.wsel)]);
    ms.window = first_i32(A.awindow[c]);
    double wm[NS], wr[NS], wmb[NS], wrb[NS];
    diag_mass_prefetch<NS>(A, row, ms, wm, wr, wmb, wrb);
    diag_mass_update<NS>(A, P, row, tid, q, var, inv_std, vard, ms, wm, wr, wmb, wrb);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        A.var[row + tid * NS + s] = var[s];
        A.inv_std[row + tid * NS + s] = inv_std[s];
    }
    if (tid == 0) {
        A.n_samples[c] = ms.n_samples;
        A.wsel[c] = ms.wsel;
        A.awindow[c] = ms.window;
        A.wsum[c * 2 + ms.wsel] = ms.wsum_f;
        A.wsum[c * 2 + (1 - ms.wsel)] = ms.wsum_b;
    }
}

// One statistic of the draws [iter_begin, iter_begin + n) of every chain out of the 64-byte records: out[c][i].
// kind 0: f64 slot `idx`; 1: i32 (0 depth / n_steps, 1 tree_size); 2: u8 flag (0 diverging, 1 tune, 2 accepted)
__global__ void stat_gather_kernel(const StatRecord* rec, long long cap, int chains, long long iter_begin, long long n, int kind,
                                   int idx, int hmc, void* out) {
    const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= static_cast<long long>(chains) * n) return;
    const long long c = k / n, i = k - c * n;
    const StatRecord& r = rec[c * cap + iter_begin + i];
    if (kind == 0) static_cast<double*>(out)[k] = r.f64[idx];
    else if (kind == 1) static_cast<int*>(out)[k] = (idx == kSiTreeSize || hmc) ? r.tree_size : static_cast<int>(r.depth_flags & 0xffffu);
    else static_cast<unsigned char*>(out)[k] = static_cast<unsigned char>((r.depth_flags >> (16 + idx)) & 1u);
}

// Streamed results (lmc_engine_copy_window_async). Both kernels WRITE THE CALLER'S ARRAYS THEMSELVES -- page-locked host memory
// mapped into the device, or device memory -- with coalesced stores; there is no staging copy and no copy-engine command per
// row (hipMemcpy2DAsync into pinned memory measured 2.8 GiB/s for 65 536 rows of 51 KB: one DMA command per row).
//
// All statistics planes of the draws [iter_begin, iter_begin + n) of every chain in ONE pass over the 64-byte records: thread
// k = (chain, draw) reads its record once and writes every plane's element in the dtype the reference's stats dict carries;
// plane p is the caller's [chains][n_out] array, the window starts at its column row0.
struct WindowPlanes {
    int n_planes;
    int kind[LMC_MAX_PLANES], idx[LMC_MAX_PLANES], as[LMC_MAX_PLANES];
    void* dst[LMC_MAX_PLANES];
};
__global__ void window_gather_kernel(const StatRecord* rec, long long cap, int chain0, int chains, long long iter_begin, long long n, int hmc,
                                     WindowPlanes W, long long n_out, long long row0) {
    const long long total = static_cast<long long>(chains) * n;   // chains [chain0, chain0 + chains) of the engine
    for (long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; k < total; k += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long c = chain0 + k / n, i = k % n;
        const StatRecord r = rec[c * cap + iter_begin + i];
        const long long o = c * n_out + row0 + i;
        for (int p = 0; p < W.n_planes; ++p) {
            void* out = W.dst[p];
            if (W.kind[p] == LMC_PLANE_F64) {
                static_cast<double*>(out)[o] = r.f64[W.idx[p]];
            } else if (W.kind[p] == LMC_PLANE_I32) {
                const int v = (W.idx[p] == kSiTreeSize || hmc) ? r.tree_size : static_cast<int>(r.depth_flags & 0xffffu);
                if (W.as[p] == LMC_AS_F64) static_cast<double*>(out)[o] = static_cast<double>(v);
                else if (W.as[p] == LMC_AS_I64) static_cast<long long*>(out)[o] = static_cast<long long>(v);
                else static_cast<int*>(out)[o] = v;
            } else {
                static_cast<unsigned char*>(out)[o] = static_cast<unsigned char>((r.depth_flags >> (16 + W.idx[p])) & 1u);
            }
        }
    }
}

// The draws of a window: for every chain `row` contiguous doubles from src + c * src_pitch to dst + c * dst_pitch. One
// workgroup walks chains blockIdx.x, blockIdx.x + gridDim.x, ...; V = 2 moves 16 bytes per lane (row and pitches even). The
// grid is small on purpose (kWindowCopyBlocks): the kernel shares the GPU with the sampling launches that follow, and a few
// hundred wavefronts of posted writes saturate the host link.
template <int V>
__global__ void window_trace_copy_kernel(const double* __restrict__ src, long long src_pitch, double* __restrict__ dst, long long dst_pitch,
                                         long long row, int chains) {
    for (int c = blockIdx.x; c < chains; c += gridDim.x) {
        const double* s = src + c * src_pitch;
        double* t = dst + c * dst_pitch;
        if constexpr (V == 2) {
            const double2* s2 = reinterpret_cast<const double2*>(s);
            double2* t2 = reinterpret_cast<double2*>(t);
            for (long long k = threadIdx.x; k < row / 2; k += blockDim.x) t2[k] = s2[k];
        } else {
            for (long long k = threadIdx.x; k < row; k += blockDim.x) t[k] = s[k];
        }
    }
}

// After lmc_engine_set_chain_state(): inv_std = 1 / sqrt(var) in float32 (quadpotential.py:226-229).
// from64: the float64 diagonal was set (QuadPotentialDiagAdapt(dtype="float64")), the float32 views follow it.
__global__ void derive_inv_std_kernel(ChainArrays A, int from64) {
    const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long n = static_cast<long long>(A.chains) * A.dpad;
    if (idx >= n) return;
    if (from64) {
        const double v = A.var64[idx];
        A.inv_std64[idx] = 1.0 / sqrt(v);
        A.var[idx] = static_cast<float>(v);
        A.inv_std[idx] = static_cast<float>(A.inv_std64[idx]);
        return;
    }
    const float sd = sqrtf(A.var[idx]);
    A.inv_std[idx] = 1.0f / sd;
    if (A.var64 != nullptr) {
        A.var64[idx] = static_cast<double>(A.var[idx]);
        A.inv_std64[idx] = static_cast<double>(A.inv_std[idx]);
    }
}

}  // namespace lmc

// ---------------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

struct lmc_engine {
    lmc_config cfg;
    int ns = 0, dpad = 0, nlds = 1, lds_bytes = 0;   // ns: vector width of the W = 1 unit kernels (dpad = 64 * ns)
    int nlds1 = 1, lds_bytes1 = 0, lds_plan = 0;      // one-wave sampling kernels: the deep-tree LDS plan (PairLds<NS, 1, 1>) and who chooses (0 / 1 pinned, 2 = per launch from the chains' reports)
    int lds_plan_wanted = 0;                          // lds_plan once a run-time compiled density has handed over its plan-1 kernel
    int plan_now = 0;                                 // the plan of the launches being enqueued (lds_plan == 2: follows the tree-size hint with hysteresis)
    int plan_last = -1;                               // the plan the most recent lmc_engine_run() actually launched with (-1: nothing launched yet)
    bool wide = false;          // the general kernels (lmc_wide.hpp): one chain = 16 wavefronts, dpad = 1024 * ns -- model_ndim > 1024,
                                // dense matrices beyond 256 dimensions, float64 adaptive diagonals
    double* init_diag64 = nullptr;   // [C][dpad] wide: the initial diagonal in float64
    int run_ns = 0, run_w = 1;                       // shape of the sampling kernel: dpad = 64 * run_ns * run_w
    hipStream_t own_stream = nullptr, stream_ = nullptr;   // stream_: use main_stream(e), which orders sub-block launches first
    // run() deals the chains to n_sub contiguous sub-blocks, each launched on its own stream: the tail of one sub-block's
    // launch is filled by the other's, and consecutive run() calls only chain up per sub-block (chains are independent)
    // (four since round 5 -- profiles/r05_sub_blocks_ab.txt: against two, C3 equal, north_star shape -0.5 %, C2 +2.4 %, C4 +1 %,
    // C5 +2.8 %; eight lose 35-40 % on C2 / C5: more streams than the hardware queues take)
#ifndef LMC_MAX_SUB
#define LMC_MAX_SUB 4      // (eight lose 35-40 %: variant builds for A/B runs pass -DLMC_MAX_SUB=8)
#endif
#ifndef LMC_DEFAULT_SUB
#define LMC_DEFAULT_SUB 4
#endif
    static constexpr int kMaxSub = LMC_MAX_SUB;
    int n_sub = 1;
    hipStream_t sub_stream[kMaxSub] = {};
    hipEvent_t sub_done[kMaxSub] = {};
    hipEvent_t main_done = nullptr;
    bool trace_external = false;   // A.trace is the caller's memory (lmc_engine_attach_trace): never freed here
    // streamed results (lmc_engine_copy_window_async): a high-priority copy stream of the engine's own, ordered after the
    // launches enqueued so far by events (the copies are kernels that write the caller's device-accessible arrays themselves)
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_dep[kMaxSub + 1] = {};
    double* chol64T = nullptr;  // FULL_F64: LT[j][i] = L[i][j] of the covariance's factor (what the state getters hand out; D.fac holds L^-1)
    uint32_t* seeds = nullptr;  // [C] the seeds of lmc_engine_seed (key of LMC_RNG_PHILOX's momentum stream)
    int* stop_host = nullptr;   // pinned, device-mapped host word the sampling kernels poll (lmc_engine_request_stop): the host
                                // sets it with a plain store -- no stream, no copy engine, no command processor in the way
    int step_jitter = 0;        // step_rand as step * uniform(lo, hi) (lmc_engine_set_step_jitter); 2: values from the host (lmc_engine_set_step_sizes)
    int step_jitter_device = 0; // what lmc_engine_set_step_jitter last asked for (0 / 1): what set_step_sizes(NULL) goes back to
    double* step_override = nullptr;   // [C] the host's step sizes for the next iteration
    double jitter_lo = 1.0, jitter_hi = 1.0;
    bool sub_pending = false;   // sub-block kernels in flight that the main stream has not been ordered after
    bool main_dirty = true;     // work enqueued on the main stream that the sub-streams have not been ordered after
    ChainArrays A;
    double* tparams = nullptr;
    int64_t n_tparams = 0;
    double* init_mean = nullptr;   // [C][dpad]
    float* init_diag = nullptr;    // [C][dpad]
    double init_weight = 10.0;
    bool potential_set = false;
    double* da_tables = nullptr;   // [2][da_table_len]: sqrt(count), count ** -k
    double initial_step = 0.0;
    // dense mass matrices (cfg.potential >= LMC_POT_FULL)
    DenseArrays D;
    int d8 = 0;                    // rows of the stored Cholesky factor (dim rounded up to 8)
    void* cov1T = nullptr;         // FULL_ADAPT: initial matrices of ONE chain (floats, or doubles with mass_f64), replicated by reset
    void* fac1 = nullptr;
    double* raw1T = nullptr;
    double* mean1 = nullptr;
    double dense_weight = 1.0, dense_multiplier = 2.0;
    int dense_window = 101, dense_update_window = 1;
    // externally evaluated density (cfg.target_family == LMC_TARGET_EXTERNAL): tick state
    TickArrays K;
    bool ticking = false;
    int* adapt_mask = nullptr;     // [C] chains whose FullAdapt.update is due after the current tick
    // run-time compiled user density (cfg.target_family == LMC_TARGET_USER in the stock library): the three kernels that
    // depend on the density functor come from a code object the caller compiled with hiprtc
    hipModule_t user_module = nullptr;
    hipFunction_t user_run = nullptr, user_trajectory = nullptr, user_logp = nullptr;
    hipFunction_t user_run1 = nullptr;               // run_kernel<NS, 1, UserTarget, 0, 1>: the sampling kernel under LDS plan 1 (optional)
    std::vector<void*> allocs;
    std::string err;
};

// Every entry point except run() works on the main stream; if run() left sub-block kernels in flight they are ordered
// before whatever comes next (event waits on the device, the host does not block).
static hipStream_t main_stream(lmc_engine* e) {
    if (e->sub_pending) {
        for (int b = 0; b < e->n_sub; ++b) {
            (void)hipEventRecord(e->sub_done[b], e->sub_stream[b]);
            (void)hipStreamWaitEvent(e->stream_, e->sub_done[b], 0);
        }
        e->sub_pending = false;
    }
    e->main_dirty = true;
    return e->stream_;
}

// run(): the sub-block streams are ordered after whatever the main stream holds. With the engine's own stream only the
// engine's entry points can have put work there (main_dirty); on a caller's stream anything may have been enqueued
// between two run() calls, so the order is established every time.
static hipError_t order_sub_blocks_after_main(lmc_engine* e) {
    if (e->n_sub <= 1) return hipSuccess;
    const bool external = e->stream_ != e->own_stream;
    if (!e->main_dirty && !external) return hipSuccess;
    hipError_t err = hipEventRecord(e->main_done, e->stream_);
    for (int b = 0; b < e->n_sub && err == hipSuccess; ++b) err = hipStreamWaitEvent(e->sub_stream[b], e->main_done, 0);
    if (err == hipSuccess) e->main_dirty = false;
    return err;
}
// ... and a caller's stream is ordered after the kernels run() just launched, so that what the caller enqueues next
// (a torch op on the trace pointer, say) sees their results -- the contract run() had when it launched on that stream.
static hipError_t order_external_stream_after_sub_blocks(lmc_engine* e) {
    if (e->n_sub <= 1 || e->stream_ == e->own_stream || !e->sub_pending) return hipSuccess;
    hipError_t err = hipSuccess;
    for (int b = 0; b < e->n_sub && err == hipSuccess; ++b) {
        err = hipEventRecord(e->sub_done[b], e->sub_stream[b]);
        if (err == hipSuccess) err = hipStreamWaitEvent(e->stream_, e->sub_done[b], 0);
    }
    if (err == hipSuccess) e->sub_pending = false;
    return err;
}

static int fail(lmc_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (e) e->err = buf;
    return code;
}

#define HIP_TRY(e, call)                                                                        \
    do {                                                                                        \
        hipError_t err__ = (call);                                                              \
        if (err__ != hipSuccess)                                                                \
            return fail((e), LMC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

template <class T>
static int dev_alloc(lmc_engine* e, T** p, size_t count, bool zero = true) {
    void* ptr = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    HIP_TRY(e, hipMalloc(&ptr, bytes));
    e->allocs.push_back(ptr);
    if (zero) HIP_TRY(e, hipMemsetAsync(ptr, 0, bytes, main_stream(e)));
    *p = static_cast<T*>(ptr);
    return LMC_OK;
}

static void dev_free(lmc_engine* e, void* p) {
    if (!p) return;
    for (auto& a : e->allocs)
        if (a == p) { a = nullptr; break; }
    (void)hipFree(p);
}

// ---- (family, NS) dispatch --------------------------------------------------------------------------
#define LMC_NS_SWITCH(e, ns, BODY)                                      \
    switch (ns) {                                                       \
        case 1: { constexpr int NS = 1; BODY; } break;                  \
        case 2: { constexpr int NS = 2; BODY; } break;                  \
        case 4: { constexpr int NS = 4; BODY; } break;                  \
        case 8: { constexpr int NS = 8; BODY; } break;                  \
        case 16: { constexpr int NS = 16; BODY; } break;                \
        default: return fail(e, LMC_ERR_INVALID, "unsupported vector width ns=%d", ns); \
    }

// M(NS, W, T) for every sampling-kernel shape of this build (LMC_PAIR_SHAPES, lmc_sampler.hpp)
#ifdef LMC_EXPERIMENTAL_SHAPES
#define LMC_FOR_EACH_SHAPE(M, T) M(1, 1, T) M(2, 1, T) M(4, 1, T) M(4, 2, T) M(4, 4, T) M(2, 8, T) M(2, 2, T) M(1, 4, T)
#else
#define LMC_FOR_EACH_SHAPE(M, T) M(1, 1, T) M(2, 1, T) M(4, 1, T) M(4, 2, T) M(4, 4, T)
#endif

#ifdef LMC_USER_TARGET_HEADER
#define LMC_USER_CASE(KERNEL_CALL) \
    case LMC_TARGET_USER: { KERNEL_CALL(UserTarget); } break;
#else
#define LMC_USER_CASE(KERNEL_CALL)
#endif

#if defined(LMC_USER_TARGET_HEADER) && defined(LMC_ONLY_USER)
// JIT build around a user density: instantiate the kernels for that family only (seconds, not a minute)
#define LMC_FAMILY_SWITCH(e, family, KERNEL_CALL)                                       \
    switch (family) {                                                                   \
        LMC_USER_CASE(KERNEL_CALL)                                                      \
        default: return fail(e, LMC_ERR_INVALID, "target family %d is not in this build", family); \
    }
#else
#define LMC_FAMILY_SWITCH(e, family, KERNEL_CALL)                                       \
    switch (family) {                                                                   \
        case LMC_TARGET_STD_NORMAL: { KERNEL_CALL(StdNormalTarget); } break;            \
        case LMC_TARGET_DIAG_GAUSSIAN: { KERNEL_CALL(DiagGaussianTarget); } break;      \
        case LMC_TARGET_AR1: { KERNEL_CALL(AR1Target); } break;                         \
        case LMC_TARGET_FUNNEL: { KERNEL_CALL(FunnelTarget); } break;                   \
        case LMC_TARGET_NORMAL1D: { KERNEL_CALL(Normal1DTarget); } break;               \
        LMC_USER_CASE(KERNEL_CALL)                                                      \
        default: return fail(e, LMC_ERR_INVALID, "unknown target family %d", family);   \
    }
#endif

template <class T>
struct DevBuf {   // RAII staging buffer: device copy of a host-or-device array
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T)); }
};

// ---- dense mass matrices: host side -----------------------------------------------------------------------
static int dense_fail(lmc_engine* e, int rc, const char* what) {
    if (rc == kDenseUnsupported)
        return fail(e, LMC_ERR_INVALID, "%s: no dense-mass kernel for target family %d / dim %d in this build", what,
                    e->cfg.target_family, e->cfg.dim);
    return fail(e, LMC_ERR_HIP, "%s: %s", what, hipGetErrorString(static_cast<hipError_t>(rc)));
}

// In-place lower Cholesky, column by column with the operation order of the device kernel (cholesky_registers in lmc_dense.hpp):
// pivot sqrt, column divided by the pivot, trailing update with one fused multiply-add per entry.
template <class T>
static bool host_cholesky(std::vector<T>& a, int d) {   // a: [d][d] row-major, lower triangle in/out
    for (int k = 0; k < d; ++k) {
        const T akk = a[static_cast<size_t>(k) * d + k];
        if (!(akk > T(0)) || !std::isfinite(akk)) return false;
        const T lkk = std::sqrt(akk);
        a[static_cast<size_t>(k) * d + k] = lkk;
        for (int i = k + 1; i < d; ++i) a[static_cast<size_t>(i) * d + k] = a[static_cast<size_t>(i) * d + k] / lkk;
        for (int j = k + 1; j < d; ++j) {
            const T ljk = a[static_cast<size_t>(j) * d + k];
            for (int i = j; i < d; ++i)
                a[static_cast<size_t>(i) * d + j] = std::fma(-a[static_cast<size_t>(i) * d + k], ljk, a[static_cast<size_t>(i) * d + j]);
        }
    }
    for (int i = 0; i < d; ++i)
        for (int j = i + 1; j < d; ++j) a[static_cast<size_t>(i) * d + j] = T(0);
    return true;
}

static int dense_reset(lmc_engine* e) {   // FULL_ADAPT: constructor state for every chain
    const int rc = dense_launch_reset(main_stream(e), e->A, e->D, e->cov1T, e->fac1, e->raw1T, e->mean1, e->dense_weight,
                                      e->dense_window, e->d8);
    if (rc != 0) return dense_fail(e, rc, "dense reset");
    return LMC_OK;
}

// potentials whose matrices live in float64 on the device (and whose momentum draw is float64)
// stop word: one chain in (mask + 1) of a launch of n chains relays the host's word (lmc_sampler.hpp: stop_request_load);
// mask + 1 = the largest power of two <= n, at most 256: every residue of (chain + git / 16) mod (mask + 1) is then owned by a
// chain that exists, so SOME chain relays at every 16th iteration whatever the launch size. (Rounded UP -- round 4 -- the
// residues n .. mask had no chain: a 130-chain launch could pass ~126 consecutive marks, ~2 000 iterations, without reading the
// host's word or writing the progress hint; found by review, tests/test_gpu_scale.py::test_interrupt_latency_of_a_130_chain_job.)
static int relay_mask_for(long long n) {
    int m = 1;
    while (2LL * m <= n && m < 256) m *= 2;
    return m - 1;
}

#ifdef LMC_USER_TARGET_HEADER
static const bool kUserCompiledIn = true;
#else
static const bool kUserCompiledIn = false;
#endif
// dynamic LDS of a sampling-kernel workgroup under plan 0 (stack + tail: MT19937, team exchange) / plan 1 (stack only)
static int sampling_lds_bytes(const lmc_engine* e, int plan = 0) {
    return plan == 1 ? e->lds_bytes1 : e->lds_bytes + lds_tail_doubles(e->run_w) * 8;
}
// Which LDS plan the launches enqueued NOW run under (lmc_sampler.hpp: run_kernel<.., PL>; results do not depend on it). Relay
// chains leave their own mean tree size of the running launch in a pinned host word as they go (stop_request_load), so
// this costs a host load -- no stream is touched, and a caller that enqueues far ahead of execution simply keeps the plan it
// started with (sample() and bench.py keep two launches in flight, so the choice follows the job). Hysteresis: up at kPlanUp
// leapfrogs per iteration, down at kPlanDown.
static int choose_lds_plan(lmc_engine* e, long long iter_begin) {
    if (e->lds_plan != 2 || e->cfg.kind != LMC_KIND_NUTS) return e->lds_plan == 1 ? 1 : 0;
    // The first 200 iterations are the regime the reference itself treats as special (early_max_treedepth, nuts.py:205-208):
    // step sizes still settle and trees are deeper than the job's own -- a launch that starts there takes plan 0 whatever is
    // reported (measured: C2 otherwise spends its third and fourth launch in the deep-tree plan, -2.4 %).
    if (iter_begin < 200) return 0;
    const int word = e->stop_host ? __atomic_load_n(e->stop_host + 24, __ATOMIC_ACQUIRE) : 0;
    const int at = word >> 12, hint = word & 4095;   // reporting iteration, mean leapfrogs per iteration of that chain's launch so far
    if (hint == 0 || at < 100) return e->plan_now;   // nothing yet / a report from inside the settling phase: keep the plan
    if (hint >= kPlanUp) e->plan_now = 1;
    else if (hint <= kPlanDown) e->plan_now = 0;
    return e->plan_now;
}

static bool pot_f64(int potential) { return potential == LMC_POT_FULL_INV || potential == LMC_POT_FULL_F64; }

#ifdef LMC_USER_TARGET_HEADER
static const bool kUserCompiledInDense = true;   // a private library around a user density: its dense kernels are the per-wave ones
#else
static const bool kUserCompiledInDense = false;
#endif
static int dense_run(lmc_engine* e, SamplerParams P) {
    P.momentum_f32 = !pot_f64(e->cfg.potential);   // quadpotential.py:452 (float32) vs :413 / dtype="float64" (float64)
    P.adapt_mass = 0;
    const bool mat_f64 = pot_f64(e->cfg.potential);
    {   // leading matrix rows each wave keeps in LDS (160 KiB per CU, allocation granule 1280 B):
        // measured at d = 128 (65 536 B matrix per chain): 0 / 16 / 32 / 48 / 64 / 96 / 128 cached rows give
        // 7.6 / 8.2 / 9.1 / 10.6 / 10.4 / 9.1 / 6.6 e7 leapfrog-steps/s -- trading waves per CU (8 -> 5) for HBM
        // bytes pays until about 5 workgroups per CU
        // are left. A matrix shared by all chains is L2 resident and its kernel VALU bound: there only the LDS
        // that costs no occupancy is used (5 workgroups per CU measured 1.36e8 against 1.75e8).
        const int by_regs = 4 * dense_waves_per_simd(e->ns);
        const int blocks_per_cu = (e->cfg.potential == LMC_POT_FULL_ADAPT && by_regs > 5) ? 5 : by_regs;
        const long budget = (163840L / blocks_per_cu) / 1280 * 1280 - dense_lds_doubles(e->dpad) * 8L;
        const long row_bytes = static_cast<long>(e->dpad) * (mat_f64 ? 8 : 4);
        long rows = budget > 0 ? budget / row_bytes : 0;
        if (e->cfg.tuning.dense_cache_rows_p1 > 0) rows = e->cfg.tuning.dense_cache_rows_p1 - 1;
        rows = rows / (2 * kSweepBatch) * (2 * kSweepBatch);
        if (rows > sweep_rows(e->cfg.dim)) rows = sweep_rows(e->cfg.dim);
        e->D.cache_rows = static_cast<int>(rows < 0 ? 0 : rows);
        // The tree's hot slots (trajectory ends, low subtree levels: level j is touched with frequency 2^-j) take
        // the LDS that is left WITHOUT lowering the workgroup count the matrix cache already implies: measured at
        // d = 32 / 64 with per-chain matrices, slots that cost occupancy lose 19 % / 24 %, free ones gain 4 - 10 %.
        const int max_levels = e->cfg.max_treedepth > e->cfg.early_max_treedepth ? e->cfg.max_treedepth : e->cfg.early_max_treedepth;
        const long used = dense_lds_doubles(e->dpad) * 8L + e->D.cache_rows * row_bytes;
        long fit = 163840L / (used > 0 ? used : 1);
        if (fit > by_regs) fit = by_regs;
        if (fit < 1) fit = 1;
        long slots = ((163840L / fit) / 1280 * 1280 - used) / (static_cast<long>(e->dpad) * 8);
        if (e->cfg.tuning.dense_lds_slots_p1 > 0) slots = e->cfg.tuning.dense_lds_slots_p1 - 1;
        if (slots > dense_scratch_vectors(max_levels)) slots = dense_scratch_vectors(max_levels);
        e->D.lds_slots = static_cast<int>(slots < 0 ? 0 : slots);
    }
    // the chains go out as sub-blocks on their own streams, like the diagonal kernels (lmc_engine_run): each sub-block is a
    // chain of launches of its own -- with FullAdapt two per tuning iteration -- and the tail of one is covered by the other
    const int n_sub = e->n_sub;
    HIP_TRY(e, order_sub_blocks_after_main(e));
    bool coop = e->cfg.potential == LMC_POT_FULL && !kUserCompiledInDense &&
                dense_coop_supported(e->cfg.target_family, e->ns, e->cfg.dim, e->dpad) != 0;
    if (e->cfg.tuning.dense_coop_off) coop = false;
    if (coop) {   // the LDS the panels and private regions leave holds the leading tree slots of the eight chains
        const int max_levels = e->cfg.max_treedepth > e->cfg.early_max_treedepth ? e->cfg.max_treedepth : e->cfg.early_max_treedepth;
        int slots = dense_coop_lds_slots(e->cfg.dim, e->dpad, dense_scratch_vectors(max_levels));
        if (e->cfg.tuning.dense_lds_slots_p1 > 0 && e->cfg.tuning.dense_lds_slots_p1 - 1 < slots) slots = e->cfg.tuning.dense_lds_slots_p1 - 1;
        e->D.cache_rows = 0;
        e->D.lds_slots = slots < 0 ? 0 : slots;
    }
    const long long end = P.iter_begin + P.n_iters;
    if (n_sub > 1) e->sub_pending = true;
    long long it = P.iter_begin;
    while (it < end) {
        // while tuning, FullAdapt refreshes covariance and factor after EVERY iteration (quadpotential.py:528-552):
        // one iteration per launch, the update kernel in between; everything else runs whole blocks of iterations
        const bool adapt = e->cfg.potential == LMC_POT_FULL_ADAPT && it < P.n_tune;
        const long long n = adapt ? 1 : end - it;
        for (int b = 0; b < n_sub; ++b) {   // (launches of the sub-blocks interleaved, so that neither stream waits for the host)
            const long long lo = static_cast<long long>(e->cfg.chains) * b / n_sub, hi = static_cast<long long>(e->cfg.chains) * (b + 1) / n_sub;
            hipStream_t st = n_sub > 1 ? e->sub_stream[b] : main_stream(e);
            SamplerParams Q = P;
            Q.iter_begin = it;
            Q.n_iters = static_cast<int>(n);
            Q.chain_begin = static_cast<int>(lo);
            Q.relay_mask = relay_mask_for(hi - lo);
            int rc;
            if (coop)   // one matrix for all chains: eight chains per workgroup, the product on the matrix cores
                rc = dense_launch_run_coop(e->cfg.target_family, e->ns, st, e->A, e->D, Q, e->tparams, static_cast<int>(hi - lo));
            else
                rc = dense_launch_run(e->cfg.target_family, e->ns, mat_f64, st, e->A, e->D, Q, e->tparams, static_cast<int>(hi - lo));
            if (rc != 0) return dense_fail(e, rc, "run");
            if (adapt) {
                rc = dense_launch_adapt(st, e->A, e->D, e->dense_multiplier, e->dense_update_window, nullptr,
                                        static_cast<int>(lo), static_cast<int>(hi - lo), static_cast<int>(it + 1));   // (skipped under a stop request unless the chain completed iteration `it`)
                if (rc != 0) return dense_fail(e, rc, "dense update");
            }
        }
        it += n;
    }
    HIP_TRY(e, order_external_stream_after_sub_blocks(e));
    return LMC_OK;
}

static int ns_for_dim(int d) {
    const int need = (d + 63) / 64;
    int ns = 1;
    while (ns < need) ns *= 2;
    return ns;
}

extern "C" {

int32_t lmc_abi_version(void) { return LMC_ABI_VERSION; }

#ifndef LMC_SOURCE_HASH
#define LMC_SOURCE_HASH "unstamped"
#endif
// the marker makes the stamp findable in the file without loading it (littlemcmc_amd/_build.py: binary_hash)
static const char kBuildStamp[] = "LMC_BUILD_HASH=" LMC_SOURCE_HASH;
const char* lmc_build_hash(void) { return kBuildStamp + 15; }

int32_t lmc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int32_t lmc_has_target(int32_t family) {
#if !(defined(LMC_USER_TARGET_HEADER) && defined(LMC_ONLY_USER))
    if (family >= LMC_TARGET_STD_NORMAL && family <= LMC_TARGET_NORMAL1D) return 1;
#endif
    if (family == LMC_TARGET_USER) return 1;       // compiled in (LMC_USER_TARGET_HEADER build) or loaded at run time
    if (family == LMC_TARGET_EXTERNAL) return 1;   // no device functor: the host evaluates the density between ticks
    return 0;
}

const char* lmc_last_error(const lmc_engine* e) {
    if (e && !e->err.empty()) return e->err.c_str();
    return g_last_error.c_str();
}

void lmc_config_defaults(lmc_config* cfg, int32_t chains, int32_t dim) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->abi_version = LMC_ABI_VERSION;
    cfg->device = 0;
    cfg->chains = chains;
    cfg->dim = dim;
    cfg->kind = LMC_KIND_NUTS;
    cfg->target_family = LMC_TARGET_STD_NORMAL;
    cfg->potential = LMC_POT_DIAG_ADAPT;
    cfg->adapt_step_size = 1;
    cfg->target_accept = 0.8;
    cfg->emax = 1000.0;
    cfg->step_scale = 0.25;
    cfg->gamma = 0.05;
    cfg->k = 0.75;
    cfg->t0 = 10.0;
    cfg->max_treedepth = 10;
    cfg->early_max_treedepth = 8;
    cfg->path_length = 2.0;
    cfg->max_steps = 1024;
    cfg->adaptation_window = 101;
    cfg->adaptation_window_multiplier = 1.0;
    cfg->rng_mode = LMC_RNG_NUMPY;
    cfg->mass_f64 = 0;
    cfg->lds_levels = 0;
    cfg->lds_plan = LMC_LDS_PLAN_AUTO;
    cfg->start_energy_sdot = LMC_SDOT_OPENBLAS_SKYLAKEX;
}

static int launch_reset(lmc_engine* e, int reset_step, int reset_mass) {
    const long long n = static_cast<long long>(e->cfg.chains) * e->dpad;
    const int threads = 256;
    const int blocks = static_cast<int>((n + threads - 1) / threads);
    const double log_step0 = std::log(e->initial_step);         // step_sizes.py:51
    const double mu = std::log(10 * e->initial_step);           // step_sizes.py:55
    LMC_LAUNCH(reset_kernel, dim3(blocks), dim3(threads), 0, main_stream(e), e->A, e->init_mean, e->init_diag,
                       e->init_diag64, e->cfg.mass_f64, e->init_weight, e->cfg.potential == LMC_POT_DIAG_ADAPT ? 1 : 0, log_step0, mu, reset_step,
                       reset_mass, e->cfg.adaptation_window);
    HIP_TRY(e, hipGetLastError());
    return LMC_OK;
}

int lmc_engine_create(const lmc_config* cfg, lmc_engine** out) {
    if (!cfg || !out) return fail(nullptr, LMC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->abi_version != LMC_ABI_VERSION)
        return fail(nullptr, LMC_ERR_INVALID, "ABI version mismatch: header %d, library %d", cfg->abi_version,
                    LMC_ABI_VERSION);
    if (cfg->chains < 1 || cfg->dim < 1) return fail(nullptr, LMC_ERR_INVALID, "chains and dim must be >= 1");
    if (cfg->lds_plan < LMC_LDS_PLAN_AUTO || cfg->lds_plan > LMC_LDS_PLAN_DEEP)
        return fail(nullptr, LMC_ERR_INVALID, "unknown lds_plan %d", cfg->lds_plan);
    if (cfg->reserved0 != 0 || cfg->tuning.reserved[0] != 0 || cfg->tuning.reserved[1] != 0 || cfg->tuning.reserved[2] != 0)
        return fail(nullptr, LMC_ERR_INVALID, "reserved fields of lmc_config must be 0");
    if (cfg->potential < LMC_POT_DIAG_ADAPT || cfg->potential > LMC_POT_FULL_F64)
        return fail(nullptr, LMC_ERR_INVALID, "unknown potential %d", cfg->potential);
    if (cfg->mass_f64 && cfg->potential > LMC_POT_DIAG && cfg->potential != LMC_POT_FULL_ADAPT)
        return fail(nullptr, LMC_ERR_INVALID, "mass_f64 is the dtype of the diagonal potentials and of LMC_POT_FULL_ADAPT "
                                              "(float64 fixed matrices: LMC_POT_FULL_F64 / LMC_POT_FULL_INV)");
    // Shapes the fused kernels are not instantiated for run in the general kernels (lmc_wide.hpp: one chain = 16 wavefronts)
    // (... and a density compiled at run time meets a dense mass matrix there: hiprtc instantiates the general kernel for it)
    const bool rtc_dense = cfg->target_family == LMC_TARGET_USER && !kUserCompiledInDense && cfg->potential >= LMC_POT_FULL &&
                           cfg->potential != LMC_POT_FULL_ADAPT;
    // cfg.tuning.force_general (a test knob like tuning.run_ns / run_w): every shape the general kernels can run takes them, so
    // that the goldens of the small shapes replay through them too
    const bool forced = cfg->tuning.force_general != 0 && cfg->rng_mode == LMC_RNG_NUMPY &&
                        (cfg->target_family != LMC_TARGET_EXTERNAL || cfg->potential < LMC_POT_FULL);
    const bool wide = cfg->dim > 1024 || (cfg->potential >= LMC_POT_FULL && cfg->dim > 256) || cfg->mass_f64 != 0 || rtc_dense || forced;
    if (wide) {
        if (cfg->dim > kWideMaxDim)
            return fail(nullptr, LMC_ERR_INVALID, "dim %d is beyond the general kernels' %d", cfg->dim, kWideMaxDim);
        if (cfg->potential >= LMC_POT_FULL && cfg->dim > kWideMaxDenseDim)
            return fail(nullptr, LMC_ERR_INVALID, "dense mass matrices are supported up to dim %d (got %d)", kWideMaxDenseDim, cfg->dim);
        if (cfg->potential == LMC_POT_FULL_ADAPT && cfg->dim > kWideMaxDenseAdaptDim)
            return fail(nullptr, LMC_ERR_INVALID, "per-chain adapted dense matrices (FULL_ADAPT) are supported up to dim %d (got %d)",
                        kWideMaxDenseAdaptDim, cfg->dim);
        if (cfg->target_family == LMC_TARGET_EXTERNAL && (cfg->potential >= LMC_POT_FULL || cfg->mass_f64))
            return fail(nullptr, LMC_ERR_INVALID, "an externally evaluated density runs with diagonal float32 mass matrices at any dim up to %d, "
                                                  "with dense ones up to dim 256; give the density as a device functor for the other shapes", kWideMaxDim);
        if (cfg->rng_mode != LMC_RNG_NUMPY)
            return fail(nullptr, LMC_ERR_INVALID, "LMC_RNG_PHILOX runs in the fused kernels only");
    }
    if (!lmc_has_target(cfg->target_family))
        return fail(nullptr, LMC_ERR_INVALID, "target family %d is not built into this library", cfg->target_family);
    if (cfg->target_family == LMC_TARGET_NORMAL1D && cfg->dim != 1)
        return fail(nullptr, LMC_ERR_INVALID, "LMC_TARGET_NORMAL1D requires dim == 1");
    if (cfg->kind != LMC_KIND_NUTS && cfg->kind != LMC_KIND_HMC)
        return fail(nullptr, LMC_ERR_INVALID, "unknown step kind %d", cfg->kind);
    if (cfg->start_energy_sdot < LMC_SDOT_NATIVE || cfg->start_energy_sdot > LMC_SDOT_OPENBLAS_HASWELL)
        return fail(nullptr, LMC_ERR_INVALID, "unknown start_energy_sdot mode %d", cfg->start_energy_sdot);
    if (cfg->adaptation_window < 1 || !(cfg->adaptation_window_multiplier > 0.0))
        return fail(nullptr, LMC_ERR_INVALID, "adaptation_window must be >= 1 and its multiplier > 0");
    if (cfg->rng_mode != LMC_RNG_NUMPY && cfg->rng_mode != LMC_RNG_PHILOX)
        return fail(nullptr, LMC_ERR_INVALID, "unknown rng_mode %d", cfg->rng_mode);
    if (cfg->rng_mode == LMC_RNG_PHILOX && (cfg->potential >= LMC_POT_FULL ||
                                            cfg->target_family == LMC_TARGET_USER || cfg->target_family == LMC_TARGET_EXTERNAL))
        return fail(nullptr, LMC_ERR_INVALID, "LMC_RNG_PHILOX runs in the fused diagonal-mass kernels with the built-in densities");
    if (cfg->max_treedepth < 1 || cfg->max_treedepth > 20 || cfg->early_max_treedepth < 1 ||
        cfg->early_max_treedepth > 20)
        return fail(nullptr, LMC_ERR_INVALID, "max_treedepth must be in [1, 20]");
    int ndev = 0;
    hipError_t herr = hipGetDeviceCount(&ndev);
    if (herr != hipSuccess || ndev < 1)
        return fail(nullptr, LMC_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(herr));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, LMC_ERR_INVALID, "device %d out of range (%d devices)", cfg->device, ndev);

    lmc_engine* e = new (std::nothrow) lmc_engine();
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "out of host memory");
    e->cfg = *cfg;
    e->wide = wide;
    e->ns = ns_for_dim(cfg->dim);
    e->dpad = 64 * e->ns;
    if (wide) {   // thread t of the chain's team owns elements t*ns .. t*ns+ns-1
        // one wavefront per chain up to 8 elements per lane (dim <= 512: 2-6x the team's rate there, tools/wide_team_ab.py),
        // else 16 wavefronts; an externally evaluated density always takes the large team (tick_wide_kernel, lmc_wide.hip, is instantiated
        // for it only). cfg.tuning.general_team = 16 is a test knob: the large team at every shape, so the small goldens replay through it too
        const bool large_team = cfg->dim > kWideOneWaveMaxDim || cfg->target_family == LMC_TARGET_EXTERNAL || cfg->tuning.general_team == 16;
        const int threads = large_team ? kWideBlock : 64;
        int wns = 1;
        while (threads * wns < cfg->dim) wns *= 2;
        e->ns = e->run_ns = wns;
        e->run_w = threads / 64;
        e->dpad = threads * wns;
    } else
    // sampling kernel: one wave per chain up to 128 elements, then 2 or 4 waves per chain
    if (e->ns <= 2) { e->run_ns = e->ns; e->run_w = 1; }
    else if (e->ns == 4) { e->run_ns = 4; e->run_w = 1; }
    else if (e->ns == 8) { e->run_ns = 4; e->run_w = 2; }
    else { e->run_ns = 4; e->run_w = 4; }
    if (!wide && cfg->tuning.run_ns > 0 && cfg->tuning.run_w > 0) {   // tuning knob: another shape of the fused kernel, 64 * ns * w == dpad
        if (64 * cfg->tuning.run_ns * cfg->tuning.run_w == e->dpad) { e->run_ns = cfg->tuning.run_ns; e->run_w = cfg->tuning.run_w; }
    }
    e->initial_step = cfg->step_scale / std::pow(static_cast<double>(cfg->dim), 0.25);   // base_hmc.py:102

    auto bail = [&](int rc) {
        lmc_engine_destroy(e);
        return rc;
    };
    hipError_t se = hipSetDevice(cfg->device);
    if (se != hipSuccess) return bail(fail(nullptr, LMC_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(se)));
    // The engine's own (main) stream is created with HIGH priority for the hardware queue that comes with it, not for the
    // priority: the runtime maps the streams of one priority onto a pool of four hardware queues, and with the main stream in
    // the same pool the fourth sub-block stream (the fifth stream) shared a queue with an earlier one -- in the first job of a
    // process its dispatch then began hundreds of iterations after the other three (d = 16), and an interrupt in that window
    // found 33 chains that had not started and returned nothing (tests/test_gpu_round5.py; LMC_SUB_BLOCKS=3 or
    // GPU_MAX_HW_QUEUES=5 made it go away, stream warm-ups did not). The main stream carries set-up and read-back work only.
    {
        int pr_low = 0, pr_high = 0;
        se = hipDeviceGetStreamPriorityRange(&pr_low, &pr_high);
        if (se == hipSuccess) se = hipStreamCreateWithPriority(&e->own_stream, hipStreamNonBlocking, pr_high);
    }
    if (se != hipSuccess) return bail(fail(nullptr, LMC_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se)));
    e->stream_ = e->own_stream;
    // sub-blocks: two halves of the chains on two streams (measured on C3's kernel: +2 % at 65 536 chains, +11 % at
    // 16 384, +22 % at 8 192, +36 % at 4 096 -- the per-launch tail of one half is covered by the other half's next launch)
    e->n_sub = (cfg->chains >= 128 && !wide) ? (LMC_DEFAULT_SUB < lmc_engine::kMaxSub ? LMC_DEFAULT_SUB : lmc_engine::kMaxSub) : 1;
    // (the dense-mass kernels keep two: their workgroups are large -- the shared-matrix kernel holds one per CU -- and a third
    //  and fourth stream's dispatch starts late enough for an interrupt to find chains that have not begun)
    if (e->n_sub > 2 && cfg->potential >= LMC_POT_FULL) e->n_sub = 2;
    if (cfg->tuning.sub_blocks >= 1 && cfg->tuning.sub_blocks <= lmc_engine::kMaxSub && cfg->tuning.sub_blocks <= cfg->chains)
        e->n_sub = cfg->tuning.sub_blocks;
    if (e->n_sub > 1) {
        for (int b = 0; b < e->n_sub; ++b) {
            se = hipStreamCreateWithFlags(&e->sub_stream[b], hipStreamNonBlocking);
            if (se == hipSuccess) se = hipEventCreateWithFlags(&e->sub_done[b], hipEventDisableTiming);
            if (se != hipSuccess) return bail(fail(nullptr, LMC_ERR_HIP, "sub-block stream: %s", hipGetErrorString(se)));
        }
        se = hipEventCreateWithFlags(&e->main_done, hipEventDisableTiming);
        if (se != hipSuccess) return bail(fail(nullptr, LMC_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(se)));
    }

    const int max_levels = (cfg->max_treedepth > cfg->early_max_treedepth ? cfg->max_treedepth
                                                                          : cfg->early_max_treedepth);
    // LDS per block: the compile-time plan PairLds<NS, W> -- reduction buffers (which double as the normals / float32-dot
    // staging area), exp table, team combine area, level scalars, cold slots, stack level 1 -- plus as many further
    // stack levels as fit WITHOUT lowering the occupancy the register budget allows (160 KiB per CU; allocation
    // granule taken as 1280 B -- measured: 12 800 B per wave keeps 12 waves/CU, 12 960 B does not); behind it the tail
    // (MT19937 state, team exchange). The rest of the stack goes to the chain's scratch row.
    int nlds = 1;
    if (wide) {
        e->nlds = 1;
        e->lds_bytes = wide_lds_bytes(e->dpad);
    } else {
        const int waves_per_cu = 4 * run_waves_per_simd(e->run_ns, e->run_w);
        const int blocks_per_cu = waves_per_cu / e->run_w > 0 ? waves_per_cu / e->run_w : 1;
        const long budget = (163840L / blocks_per_cu) / 1280 * 1280 - lds_tail_doubles(e->run_w) * 8L;
        if (pair_min_doubles(e->run_ns, e->run_w) * 8L > budget)
            return bail(fail(nullptr, LMC_ERR_INVALID, "the sampling kernel's LDS plan does not fit the budget"));
        nlds = cfg->lds_levels > 0 ? cfg->lds_levels : 1;
        if (cfg->lds_levels <= 0)
            while (nlds < max_levels && pair_total_doubles(e->run_ns, e->run_w, nlds + 1) * 8L <= budget) ++nlds;
        if (nlds > max_levels) nlds = max_levels;
        if (nlds < 1) nlds = 1;
        e->nlds = nlds;
        e->lds_bytes = pair_total_doubles(e->run_ns, e->run_w, nlds) * 8;
        // The deep-tree plan of the one-wave kernels (lmc_sampler.hpp: PairLds<NS, 1, 1>, run_kernel's kDynPlan): MT19937 used in
        // place, one cold slot (none at NS = 4) in LDS, and the room that frees holds stack level 2. Same budget per wave (the
        // generator's 2.5 KB included, since it is not in LDS under this plan). By default the engine picks the plan of every
        // launch it enqueues from the tree sizes the running chains report (choose_lds_plan); cfg.lds_plan pins it (A/B runs, the
        // bit-identity test, either kernel under the oracle), and an explicit cfg.lds_levels (a test knob for plan 0's level
        // count) pins plan 0.
        e->nlds1 = nlds;
        e->lds_bytes1 = 0;
        e->lds_plan = 0;
        const long budget1 = (163840L / blocks_per_cu) / 1280 * 1280;
        if (e->run_w == 1 && run_mt_in_lds(1) && cfg->lds_levels <= 0 && pair_min_doubles(e->run_ns, 1, 1) * 8L <= budget1) {
            int n1 = 1;
            while (n1 < max_levels && pair_total_doubles(e->run_ns, 1, n1 + 1, 1) * 8L <= budget1) ++n1;
            e->nlds1 = n1;
            e->lds_bytes1 = pair_total_doubles(e->run_ns, 1, n1, 1) * 8;
            e->lds_plan = cfg->lds_plan == LMC_LDS_PLAN_SHALLOW ? 0 : cfg->lds_plan == LMC_LDS_PLAN_DEEP ? 1 : 2;
            if (e->nlds1 <= e->nlds) e->lds_plan = 0;   // nothing gained: the plan would only move the generator out
            if (cfg->rng_mode == LMC_RNG_PHILOX) e->lds_plan = 0;   // (instantiated on the parity stream)
            e->lds_plan_wanted = e->lds_plan;
            if (cfg->target_family == LMC_TARGET_USER && !kUserCompiledIn)
                e->lds_plan = 0;                         // until lmc_engine_load_user_run_plan1() hands the plan-1 kernel over
            e->plan_now = e->lds_plan == 1 ? 1 : 0;
        }
    }
    if (e->lds_bytes > 160 * 1024) return bail(fail(nullptr, LMC_ERR_INVALID, "lds_levels too large"));

    const size_t C = cfg->chains, dp = e->dpad;
    ChainArrays& A = e->A;
    std::memset(&A, 0, sizeof(A));
    A.chains = cfg->chains;
    A.d = cfg->dim;
    A.dpad = e->dpad;
    int rc = LMC_OK;
#define TRY_ALLOC(x) if ((rc = (x)) != LMC_OK) return bail(rc)
    TRY_ALLOC(dev_alloc(e, &A.q, C * dp));
    TRY_ALLOC(dev_alloc(e, &A.var, C * dp));
    TRY_ALLOC(dev_alloc(e, &A.inv_std, C * dp));
    TRY_ALLOC(dev_alloc(e, &A.wmean, 2 * C * dp));
    TRY_ALLOC(dev_alloc(e, &A.wraw, 2 * C * dp));
    TRY_ALLOC(dev_alloc(e, &A.wsum, C * 2));
    TRY_ALLOC(dev_alloc(e, &A.wsel, C));
    TRY_ALLOC(dev_alloc(e, &A.n_samples, C));
    TRY_ALLOC(dev_alloc(e, &A.awindow, C));
    TRY_ALLOC(dev_alloc(e, &A.da, C * 4));
    TRY_ALLOC(dev_alloc(e, &A.da_count, C));
    TRY_ALLOC(dev_alloc(e, &A.iter_count, C));
    TRY_ALLOC(dev_alloc(e, &A.mt, C * kMtN));
    TRY_ALLOC(dev_alloc(e, &A.rng_pos, C));
    TRY_ALLOC(dev_alloc(e, &A.rng_has_gauss, C));
    TRY_ALLOC(dev_alloc(e, &A.rng_gauss, C));
    TRY_ALLOC(dev_alloc(e, &A.status, C));
    TRY_ALLOC(dev_alloc(e, &A.counters, C * kNumCounters));
    {   // the stop word lives in pinned host memory mapped into the device (uncached, coherent): a request is a host store
        void* dev_view = nullptr;
        // (two words a cache line apart: the stop request the host writes and the device reads, and the progress hint the
        //  device writes and the host reads)
        if (hipHostMalloc(reinterpret_cast<void**>(&e->stop_host), 128, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer(&dev_view, e->stop_host, 0) != hipSuccess)
            return bail(fail(nullptr, LMC_ERR_HIP, "stop word: %s", hipGetErrorString(hipGetLastError())));
        std::memset(e->stop_host, 0, 128);
        A.stop = static_cast<const int*>(dev_view);
        A.progress = static_cast<int*>(dev_view) + 16;
        A.tree_hint = static_cast<int*>(dev_view) + 24;   // mean tree size a relay chain reports (choose_lds_plan); 0 = nothing yet
    }
    TRY_ALLOC(dev_alloc(e, &A.stop_dev, 1));   // what the relay chains set and every chain reads (zeroed)
    TRY_ALLOC(dev_alloc(e, &e->seeds, C));
    A.seed = e->seeds;
    A.scratch_stride = static_cast<long long>(max_levels - nlds + 1) * 4 * dp + static_cast<long long>(kNumColdSlots) * dp;
    const bool dense = cfg->potential >= LMC_POT_FULL;
    if (dense) A.scratch_stride = static_cast<long long>(dense_scratch_vectors(max_levels)) * dp;
    const bool external = cfg->target_family == LMC_TARGET_EXTERNAL;
    if (external) A.scratch_stride = static_cast<long long>(dense ? tick_dense_scratch_vectors(max_levels) : tick_scratch_vectors(max_levels)) * dp;
    if (wide) {
        if (!external) A.scratch_stride = static_cast<long long>(wide_scratch_slots(max_levels)) * dp;
        TRY_ALLOC(dev_alloc(e, &A.var64, C * dp));
        TRY_ALLOC(dev_alloc(e, &A.inv_std64, C * dp));
        TRY_ALLOC(dev_alloc(e, &e->init_diag64, C * dp));
    }
    TRY_ALLOC(dev_alloc(e, &A.scratch, C * static_cast<size_t>(A.scratch_stride), false));
    TRY_ALLOC(dev_alloc(e, &e->init_mean, C * dp));
    TRY_ALLOC(dev_alloc(e, &e->init_diag, C * dp));
    TRY_ALLOC(dev_alloc(e, &e->tparams, 8));
    {   // dual-averaging tables: sqrt(count), count ** -k with the HOST libm (step_sizes.py:88-89)
        const int len = 16384;
        std::vector<double> tab(2 * static_cast<size_t>(len));
        for (int i = 0; i < len; ++i) {
            tab[i] = std::sqrt(static_cast<double>(i));
            tab[len + i] = i > 0 ? std::pow(static_cast<double>(i), -cfg->k) : 0.0;
        }
        TRY_ALLOC(dev_alloc(e, &e->da_tables, tab.size(), false));
        hipError_t ce = hipMemcpy(e->da_tables, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
        if (ce != hipSuccess) return bail(fail(nullptr, LMC_ERR_HIP, "da tables: %s", hipGetErrorString(ce)));
        A.da_sqrt = e->da_tables;
        A.da_mk = e->da_tables + len;
        A.da_table_len = len;
    }
#undef TRY_ALLOC
    std::memset(&e->K, 0, sizeof(e->K));
    if (external) {
        TickArrays& K = e->K;
        if ((rc = dev_alloc(e, &K.phase, C)) != LMC_OK) return bail(rc);
        if ((rc = dev_alloc(e, &K.git, C)) != LMC_OK) return bail(rc);
        if ((rc = dev_alloc(e, &K.ti, C * kNumTickInt)) != LMC_OK) return bail(rc);
        if ((rc = dev_alloc(e, &K.td, C * kNumTickDbl)) != LMC_OK) return bail(rc);
        if ((rc = dev_alloc(e, &K.lvl, C * 4 * kTickLevels)) != LMC_OK) return bail(rc);
        if ((rc = dev_alloc(e, &K.q_eval, C * static_cast<size_t>(cfg->dim))) != LMC_OK) return bail(rc);
        if ((rc = dev_alloc(e, &K.n_active, 1)) != LMC_OK) return bail(rc);
        if ((rc = dev_alloc(e, &e->adapt_mask, C)) != LMC_OK) return bail(rc);
    }
    std::memset(&e->D, 0, sizeof(e->D));
    if (dense) {
        DenseArrays& D = e->D;
        const size_t d = cfg->dim;
        e->d8 = (cfg->dim + 7) / 8 * 8;
        D.kind = cfg->potential == LMC_POT_FULL_F64 ? static_cast<int>(kDenseFullInv) : cfg->potential;   // device code: float64 sweeps, momentum = sweep of D.fac
        const bool per_chain = cfg->potential == LMC_POT_FULL_ADAPT;
        const size_t P = per_chain ? C : 1;
        const size_t drows = sweep_rows(cfg->dim);
        D.mat_stride = per_chain ? static_cast<long long>(drows * dp) : 0;
        D.fac_stride = per_chain ? static_cast<long long>(e->d8) * static_cast<long long>(dp) : 0;
        const bool adapt_f64 = per_chain && cfg->mass_f64 != 0;   // QuadPotentialFullAdapt(dtype="float64")
        D.mat_f64 = (pot_f64(cfg->potential) || adapt_f64) ? 1 : 0;
        if (pot_f64(cfg->potential)) {
            double *m = nullptr, *f = nullptr;
            if ((rc = dev_alloc(e, &m, drows * dp)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &f, drows * dp)) != LMC_OK) return bail(rc);
            D.covT = m; D.fac = f;
            if (cfg->potential == LMC_POT_FULL_F64 && (rc = dev_alloc(e, &e->chol64T, drows * dp)) != LMC_OK) return bail(rc);
        } else if (adapt_f64) {
            double *m = nullptr, *f = nullptr, *w = nullptr;
            if ((rc = dev_alloc(e, &m, P * drows * dp)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &f, P * e->d8 * dp)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &w, P * drows * dp)) != LMC_OK) return bail(rc);   // the float64 refresh always factorises through HBM
            D.covT = m; D.fac = f; D.chol_work = w;
        } else {
            float *m = nullptr, *f = nullptr;
            if ((rc = dev_alloc(e, &m, P * drows * dp)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &f, P * e->d8 * dp)) != LMC_OK) return bail(rc);
            D.covT = m; D.fac = f;
            if (!per_chain) {
                double* fi = nullptr;
                if ((rc = dev_alloc(e, &fi, drows * dp)) != LMC_OK) return bail(rc);
                D.fac_inv = fi;
            }
        }
        if (per_chain) {
            if ((rc = dev_alloc(e, &D.rawT, 2 * C * d * dp)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &D.emean, 2 * C * dp)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &D.en, 2 * C)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &D.esel, C)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &D.prev_update, C)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &D.window, C)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &D.chol_failed, C)) != LMC_OK) return bail(rc);
            if (adapt_f64) {
                double *c1 = nullptr, *f1 = nullptr;
                if ((rc = dev_alloc(e, &c1, d * dp)) != LMC_OK) return bail(rc);
                if ((rc = dev_alloc(e, &f1, static_cast<size_t>(e->d8) * dp)) != LMC_OK) return bail(rc);
                e->cov1T = c1; e->fac1 = f1;
            } else {
                float *c1 = nullptr, *f1 = nullptr;
                if ((rc = dev_alloc(e, &c1, d * dp)) != LMC_OK) return bail(rc);
                if ((rc = dev_alloc(e, &f1, static_cast<size_t>(e->d8) * dp)) != LMC_OK) return bail(rc);
                e->cov1T = c1; e->fac1 = f1;
            }
            if ((rc = dev_alloc(e, &e->raw1T, d * dp)) != LMC_OK) return bail(rc);
            if ((rc = dev_alloc(e, &e->mean1, dp)) != LMC_OK) return bail(rc);
            // beyond 256 dimensions the refresh factorises through HBM (lmc_dense.hpp: cholesky_hbm) and needs a work area per
            // chain; cfg.tuning.chol_hbm (a test knob) gives the small shapes one too, and dense_launch_adapt then takes that form
            D.force_chol_hbm = cfg->tuning.chol_hbm != 0 ? 1 : 0;
            if (!adapt_f64 && (cfg->dim > kDenseAdaptRegisterMaxDim || D.force_chol_hbm)) {
                float* w = nullptr;
                if ((rc = dev_alloc(e, &w, C * drows * dp)) != LMC_OK) return bail(rc);
                D.chol_work = w;
            }
        }
    }
    // default potential of BaseHMC (base_hmc.py:109-113): QuadPotentialDiagAdapt(d, zeros, ones, 10); a dense
    // engine starts from the identity (QuadPotentialFullAdapt's own default, quadpotential.py:501-503)
    {
        std::vector<double> ones(cfg->dim, 1.0), zeros(cfg->dim, 0.0);
        rc = lmc_engine_set_potential(e, zeros.data(), ones.data(), 10.0, 0);
        if (rc != LMC_OK) return bail(rc);
        rc = launch_reset(e, 1, 0);   // DualAverageAdaptation.__init__ -> reset() (step_sizes.py:41-56), iter_count = 0
        if (rc != LMC_OK) return bail(rc);
        if (dense) {
            std::vector<double> eye(static_cast<size_t>(cfg->dim) * cfg->dim, 0.0);
            for (int i = 0; i < cfg->dim; ++i) eye[static_cast<size_t>(i) * cfg->dim + i] = 1.0;
            rc = lmc_engine_set_dense_potential(e, eye.data(), zeros.data(), 1.0, 101, 2.0, 1);
            if (rc != LMC_OK) return bail(rc);
        }
        // seed 0 so that an engine used without lmc_engine_seed() is still deterministic
        std::vector<uint32_t> seeds(C, 0u);
        rc = lmc_engine_seed(e, seeds.data());
        if (rc != LMC_OK) return bail(rc);
    }
    se = hipStreamSynchronize(main_stream(e));
    if (se != hipSuccess) return bail(fail(nullptr, LMC_ERR_HIP, "engine init: %s", hipGetErrorString(se)));
    *out = e;
    return LMC_OK;
}

void lmc_engine_destroy(lmc_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    if (e->stream_) (void)hipStreamSynchronize(main_stream(e));
    for (int b = 0; b < lmc_engine::kMaxSub; ++b) {
        if (e->sub_stream[b]) { (void)hipStreamSynchronize(e->sub_stream[b]); (void)hipStreamDestroy(e->sub_stream[b]); }
        if (e->sub_done[b]) (void)hipEventDestroy(e->sub_done[b]);
    }
    if (e->main_done) (void)hipEventDestroy(e->main_done);
    if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
    for (hipEvent_t ev : e->copy_dep)
        if (ev) (void)hipEventDestroy(ev);
    if (e->stop_host) (void)hipHostFree(e->stop_host);
    for (void* p : e->allocs)
        if (p) (void)hipFree(p);
    if (e->user_module) (void)hipModuleUnload(e->user_module);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

// ---- run-time compiled user density ------------------------------------------------------------------------------

int lmc_engine_kernel_shape(lmc_engine* e, int32_t* unit_ns, int32_t* run_ns, int32_t* run_w) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    if (unit_ns) *unit_ns = e->ns;
    if (run_ns) *run_ns = e->run_ns;
    if (run_w) *run_w = e->run_w;
    return LMC_OK;
}

int32_t lmc_engine_uses_general_kernels(lmc_engine* e) { return (e && e->wide) ? 1 : 0; }

int lmc_engine_occupancy(lmc_engine* e, int32_t* resident_chains, int32_t* waves_per_chain, double* wall_clock_hz) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (waves_per_chain) *waves_per_chain = e->run_w;
    if (wall_clock_hz) {
        int khz = 0;
        HIP_TRY(e, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->cfg.device));
        *wall_clock_hz = 1e3 * khz;
    }
    if (!resident_chains) return LMC_OK;
    *resident_chains = 0;
    if (e->cfg.target_family == LMC_TARGET_EXTERNAL || e->cfg.potential >= LMC_POT_FULL || e->wide) return LMC_OK;
    int cus = 0;
    HIP_TRY(e, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->cfg.device));
    const int run_lds = sampling_lds_bytes(e);
    const int block = 64 * e->run_w;
    int per_cu = 0;
    if (e->cfg.target_family == LMC_TARGET_USER && !kUserCompiledIn) {
        if (!e->user_run) return fail(e, LMC_ERR_STATE, "the user density's kernels are not loaded");
        HIP_TRY(e, hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, e->user_run, block, static_cast<size_t>(run_lds)));
        *resident_chains = per_cu * cus;
        return LMC_OK;
    }
    // (the very instantiation run() launches: the counter-based momentum stream is its own kernel with its own registers)
#define OCC_ONE(NSV, WV, T)                                                                                    \
    if (!found && e->run_ns == NSV && e->run_w == WV) {                                                        \
        found = true;                                                                                          \
        if (e->cfg.rng_mode == LMC_RNG_PHILOX) {                                                               \
            if (run_lds > 64 * 1024)                                                                           \
                HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&run_kernel<NSV, WV, T, 1>),      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, run_lds));          \
            HIP_TRY(e, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, run_kernel<NSV, WV, T, 1>, block, \
                                                                    static_cast<size_t>(run_lds)));            \
        } else {                                                                                               \
            if (run_lds > 64 * 1024)                                                                           \
                HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&run_kernel<NSV, WV, T>),         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, run_lds));          \
            HIP_TRY(e, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, run_kernel<NSV, WV, T>, block,    \
                                                                    static_cast<size_t>(run_lds)));            \
        }                                                                                                      \
    }
    // every shape of LMC_PAIR_SHAPES (lmc_sampler.hpp), the experimental ones of a variant build included
#define OCC_CALL(T)                                                                                            \
    {                                                                                                          \
        bool found = false;                                                                                    \
        LMC_FOR_EACH_SHAPE(OCC_ONE, T)                                                                         \
        if (!found) return fail(e, LMC_ERR_INVALID, "unsupported kernel shape ns=%d w=%d", e->run_ns, e->run_w); \
    }
    LMC_FAMILY_SWITCH(e, e->cfg.target_family, OCC_CALL)
#undef OCC_CALL
#undef OCC_ONE
    *resident_chains = per_cu * cus;
    return LMC_OK;
}

int32_t lmc_engine_run_lds_bytes(lmc_engine* e) {
    if (!e || e->cfg.target_family == LMC_TARGET_EXTERNAL || (e->cfg.potential >= LMC_POT_FULL && !e->wide)) return -1;
    if (e->wide) return e->lds_bytes;
    return sampling_lds_bytes(e, e->plan_last >= 0 ? e->plan_last : e->plan_now);   // the launch that ran, not the one that may come
}

int32_t lmc_engine_last_run_plan(lmc_engine* e) {
    if (!e || e->plan_last < 0 || e->wide || e->cfg.potential >= LMC_POT_FULL || e->cfg.target_family == LMC_TARGET_EXTERNAL) return 0;
    return e->plan_last == 1 ? LMC_LDS_PLAN_DEEP : LMC_LDS_PLAN_SHALLOW;
}

int lmc_engine_load_user_kernels(lmc_engine* e, const void* code_object, const char* run_name, const char* trajectory_name,
                                 const char* logp_name) {
    if (!e || !code_object || !run_name || !trajectory_name || !logp_name) return fail(e, LMC_ERR_INVALID, "null argument");
    if (e->cfg.target_family != LMC_TARGET_USER)
        return fail(e, LMC_ERR_STATE, "lmc_engine_load_user_kernels() needs cfg.target_family = LMC_TARGET_USER");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    if (e->user_module) { (void)hipModuleUnload(e->user_module); e->user_module = nullptr; }
    e->user_run = e->user_trajectory = e->user_logp = nullptr;
    e->user_run1 = nullptr;
    if (e->cfg.target_family == LMC_TARGET_USER && !kUserCompiledIn) { e->lds_plan = 0; e->plan_now = 0; }   // plan 0 until the plan-1 kernel is handed over
    HIP_TRY(e, hipModuleLoadData(&e->user_module, code_object));
    HIP_TRY(e, hipModuleGetFunction(&e->user_run, e->user_module, run_name));
    HIP_TRY(e, hipModuleGetFunction(&e->user_trajectory, e->user_module, trajectory_name));
    HIP_TRY(e, hipModuleGetFunction(&e->user_logp, e->user_module, logp_name));
    return LMC_OK;
}

int lmc_engine_load_user_run_plan1(lmc_engine* e, const char* run_name_plan1) {
    if (!e || !run_name_plan1) return fail(e, LMC_ERR_INVALID, "null argument");
    if (!e->user_module) return fail(e, LMC_ERR_STATE, "lmc_engine_load_user_kernels() first");
    if (e->wide || e->run_w != 1) return fail(e, LMC_ERR_INVALID, "the deep-tree LDS plan exists for the one-wave sampling kernels");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipModuleGetFunction(&e->user_run1, e->user_module, run_name_plan1));
    e->lds_plan = e->lds_plan_wanted;
    e->plan_now = e->lds_plan == 1 ? 1 : 0;
    return LMC_OK;
}

// launch one of the module's kernels: the arguments are the very values the compiled-in kernels take
static int user_launch(lmc_engine* e, hipFunction_t f, hipStream_t st, unsigned grid, unsigned block, unsigned lds, void** args) {
    if (!f)
        return fail(e, LMC_ERR_STATE, "the user density's kernels are not loaded: call lmc_engine_load_user_kernels() "
                                      "(littlemcmc_amd.targets.UserTarget does)");
    if (e->cfg.potential >= LMC_POT_FULL && !e->wide)
        return fail(e, LMC_ERR_INVALID, "a run-time compiled user density runs with diagonal mass matrices in the fused kernels; "
                                        "dense ones (Full / FullInv) take the general kernels, FullAdapt needs the density compiled in (UserTarget(..., jit=\"hipcc\"))");
    (void)hipGetLastError();
    HIP_TRY(e, hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, lds, st, args, nullptr));
    return LMC_OK;
}

int lmc_engine_set_step_sizes(lmc_engine* e, const double* step_sizes) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (!step_sizes) {   // back to the adapted step sizes (or the device's own jitter, if that was set before)
        if (e->step_jitter == 2) e->step_jitter = e->step_jitter_device;
        return LMC_OK;
    }
    if (!e->step_override) {
        int rc = dev_alloc(e, &e->step_override, static_cast<size_t>(e->cfg.chains));
        if (rc != LMC_OK) return rc;
        e->A.step_override = e->step_override;
    }
    HIP_TRY(e, hipMemcpyAsync(e->step_override, step_sizes, sizeof(double) * e->cfg.chains, hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));   // the caller's array may be reused at once
    e->step_jitter = 2;
    return LMC_OK;
}

int lmc_engine_diag_update(lmc_engine* e, int32_t tune) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    if (e->cfg.potential != LMC_POT_DIAG_ADAPT) return LMC_OK;   // QuadPotential.update of the fixed potentials: `pass` (quadpotential.py:112-118)
    if (!tune) return LMC_OK;                                    // quadpotential.py:233-234
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    SamplerParams P;
    std::memset(&P, 0, sizeof(P));
    P.window = e->cfg.adaptation_window;
    P.window_multiplier = e->cfg.adaptation_window_multiplier;
    P.mass_f64 = e->cfg.mass_f64 ? 1 : 0;
    if (e->wide) {
        const int rc = wide_launch_mass_update(e->ns, e->run_w, main_stream(e), e->A, P);
        if (rc != 0) return dense_fail(e, rc, "diag_update (general kernel)");
        return LMC_OK;
    }
    const dim3 grid(e->cfg.chains), block(64);
    LMC_NS_SWITCH(e, e->ns, LMC_LAUNCH((mass_update_kernel<NS>), grid, block, 0, main_stream(e), e->A, P))
    HIP_TRY(e, hipGetLastError());
    return LMC_OK;
}

int lmc_engine_set_step_jitter(lmc_engine* e, int32_t enable, double lo, double hi) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    if (enable && !(std::isfinite(lo) && std::isfinite(hi))) return fail(e, LMC_ERR_INVALID, "step jitter bounds must be finite");
    e->step_jitter_device = enable ? 1 : 0;
    if (e->step_jitter != 2) e->step_jitter = e->step_jitter_device;   // (a host override in force stays in force until set_step_sizes(NULL))
    e->jitter_lo = lo;
    e->jitter_hi = hi;
    return LMC_OK;
}

// Ctrl-C: every chain leaves its launch within a few iterations; launches still queued return at once. The word is a
// host store into pinned, device-mapped memory; every 256th chain of a launch reads it every 16th iteration and relays it
// to a device word that all chains look at once per iteration (lmc_sampler.hpp: stop_request_load). A device word set through a stream (hipStreamWriteValue32 / hipMemcpyAsync on a
// stream of its own) was tried first: with every wave slot held by sampling kernels the command processor delivered it
// only after 0.1 s (short waves) to seconds (waves that live as long as the launch), i.e. not at all for whole-job
// launches (tools/ubench/stop_probe.hip).
int lmc_engine_request_stop(lmc_engine* e, int32_t stop) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (stop) {
        __atomic_store_n(e->stop_host, 1, __ATOMIC_RELEASE);
    } else {   // re-armed in order: after everything that was launched under the request has drained
        hipStream_t st = main_stream(e);
        HIP_TRY(e, hipStreamSynchronize(st));
        __atomic_store_n(e->stop_host, 0, __ATOMIC_RELEASE);
        HIP_TRY(e, hipMemsetAsync(e->A.stop_dev, 0, sizeof(int), st));
        HIP_TRY(e, hipStreamSynchronize(st));
    }
    return LMC_OK;
}

// Where the job is, without touching a stream: the iteration index a relay chain of the running launch last started
// (written into pinned host memory every 16th iteration, next to reading the stop word). A hint -- chains advance at their
// own pace and relays take turns -- for progress lines and callbacks; what every chain has COMPLETED is iter_count
// (lmc_engine_get_chain_state), which waits for the launches.
int64_t lmc_engine_progress(lmc_engine* e) {
    if (!e || !e->stop_host) return 0;
    return static_cast<int64_t>(__atomic_load_n(e->stop_host + 16, __ATOMIC_ACQUIRE));
}

int lmc_engine_set_stream(lmc_engine* e, void* hip_stream) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    e->stream_ = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->own_stream;
    e->main_dirty = true;
    return LMC_OK;
}

int lmc_engine_synchronize(lmc_engine* e) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

int lmc_engine_set_target_params(lmc_engine* e, const double* params, int64_t n) {
    if (!e || n < 0 || (n > 0 && !params)) return fail(e, LMC_ERR_INVALID, "bad target params");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (e->cfg.target_family == LMC_TARGET_DIAG_GAUSSIAN && n != e->cfg.dim)
        return fail(e, LMC_ERR_INVALID, "diag_gaussian needs %d precisions, got %lld", e->cfg.dim, (long long)n);
    if (e->cfg.target_family == LMC_TARGET_AR1 && n != 3)
        return fail(e, LMC_ERR_INVALID, "ar1 needs params {c_end, c_mid, off}");
    if (e->cfg.target_family == LMC_TARGET_NORMAL1D && n != 2)
        return fail(e, LMC_ERR_INVALID, "normal1d needs params {loc, scale}");
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    dev_free(e, e->tparams);
    e->tparams = nullptr;
    int rc = dev_alloc(e, &e->tparams, static_cast<size_t>(n > 8 ? n : 8));
    if (rc != LMC_OK) return rc;
    if (n > 0) HIP_TRY(e, hipMemcpyAsync(e->tparams, params, n * sizeof(double), hipMemcpyDefault, main_stream(e)));
    e->n_tparams = n;
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

int lmc_engine_set_potential(lmc_engine* e, const double* initial_mean, const double* initial_diag,
                             double initial_weight, int32_t per_chain) {
    if (!e || !initial_diag) return fail(e, LMC_ERR_INVALID, "initial_diag is required");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const int C = e->cfg.chains, d = e->cfg.dim, dp = e->dpad;
    // stage on the host: float32 cast of the diagonal (quadpotential.py:181-183 / :359) and padding
    std::vector<double> hdiag(static_cast<size_t>(per_chain ? C : 1) * d), hmean(hdiag.size(), 0.0);
    HIP_TRY(e, hipMemcpy(hdiag.data(), initial_diag, hdiag.size() * sizeof(double), hipMemcpyDefault));
    if (initial_mean)
        HIP_TRY(e, hipMemcpy(hmean.data(), initial_mean, hmean.size() * sizeof(double), hipMemcpyDefault));
    for (size_t i = 0; i < hdiag.size(); ++i)
        if (!(hdiag[i] > 0.0))   // partial_check_positive_definite (quadpotential.py:68-77)
            return fail(e, LMC_ERR_INVALID, "Scaling is not positive definite: diagonal entry %zu is %g", i, hdiag[i]);
    std::vector<float> fdiag(static_cast<size_t>(C) * dp, 1.0f);
    std::vector<double> fmean(static_cast<size_t>(C) * dp, 0.0), ddiag(e->wide ? static_cast<size_t>(C) * dp : 0, 1.0);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < d; ++i) {
            const size_t src = static_cast<size_t>(per_chain ? c : 0) * d + i;
            fdiag[static_cast<size_t>(c) * dp + i] = static_cast<float>(hdiag[src]);
            fmean[static_cast<size_t>(c) * dp + i] = hmean[src];
            if (e->wide)   // initial_diag.astype(dtype) (quadpotential.py:181-183)
                ddiag[static_cast<size_t>(c) * dp + i] = e->cfg.mass_f64 ? hdiag[src] : static_cast<double>(static_cast<float>(hdiag[src]));
        }
    if (e->wide)
        HIP_TRY(e, hipMemcpyAsync(e->init_diag64, ddiag.data(), ddiag.size() * sizeof(double), hipMemcpyHostToDevice, main_stream(e)));
    HIP_TRY(e, hipMemcpyAsync(e->init_diag, fdiag.data(), fdiag.size() * sizeof(float), hipMemcpyHostToDevice, main_stream(e)));
    HIP_TRY(e, hipMemcpyAsync(e->init_mean, fmean.data(), fmean.size() * sizeof(double), hipMemcpyHostToDevice, main_stream(e)));
    e->init_weight = initial_weight;
    e->potential_set = true;
    // only the mass state: the reference's potential.reset() / BaseHMC.reset() leave the step-size adaptation alone
    // (base_hmc.py:196-200); create() and reset_tuning() initialise the dual averaging
    int rc = launch_reset(e, 0, 1);
    if (rc != LMC_OK) return rc;
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));   // host staging buffers go out of scope
    return LMC_OK;
}

int lmc_engine_set_dense_potential(lmc_engine* e, const double* matrix, const double* initial_mean,
                                   double initial_weight, int32_t adaptation_window,
                                   double adaptation_window_multiplier, int32_t update_window) {
    if (!e || !matrix) return fail(e, LMC_ERR_INVALID, "matrix is required");
    if (e->cfg.potential < LMC_POT_FULL)
        return fail(e, LMC_ERR_STATE, "the engine was created with a diagonal potential (cfg.potential = %d)", e->cfg.potential);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    const int d = e->cfg.dim, dp = e->dpad, d8 = e->d8;
    const size_t dd = static_cast<size_t>(d) * d;
    std::vector<double> m(dd), mean(d, 0.0);
    HIP_TRY(e, hipMemcpy(m.data(), matrix, dd * sizeof(double), hipMemcpyDefault));
    if (initial_mean) HIP_TRY(e, hipMemcpy(mean.data(), initial_mean, d * sizeof(double), hipMemcpyDefault));
    for (size_t i = 0; i < dd; ++i)
        if (!std::isfinite(m[i])) return fail(e, LMC_ERR_INVALID, "array must not contain infs or NaNs");
    if (e->cfg.potential == LMC_POT_FULL_INV) {
        // L = cholesky(A) (quadpotential.py:402); velocity = cho_solve((L, True), x) (:406) is served by the
        // explicit inverse  Sigma = L^-T L^-1  (formed once, extended precision), random = L n (:413)
        std::vector<double> L(m);
        if (!host_cholesky(L, d)) return fail(e, LMC_ERR_INVALID, "matrix is not positive definite");
        std::vector<long double> Li(dd, 0.0L);   // L^-1, lower
        for (int c = 0; c < d; ++c)
            for (int i = c; i < d; ++i) {
                long double acc = (i == c) ? 1.0L : 0.0L;
                for (int k = c; k < i; ++k) acc -= static_cast<long double>(L[static_cast<size_t>(i) * d + k]) * Li[static_cast<size_t>(k) * d + c];
                Li[static_cast<size_t>(i) * d + c] = acc / L[static_cast<size_t>(i) * d + i];
            }
        std::vector<double> covT(static_cast<size_t>(d) * dp, 0.0), LT(static_cast<size_t>(d) * dp, 0.0);
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                long double acc = 0.0L;
                for (int k = (i > j ? i : j); k < d; ++k) acc += Li[static_cast<size_t>(k) * d + i] * Li[static_cast<size_t>(k) * d + j];
                covT[static_cast<size_t>(j) * dp + i] = static_cast<double>(acc);           // Sigma[i][j]
                LT[static_cast<size_t>(j) * dp + i] = L[static_cast<size_t>(i) * d + j];    // L[i][j]
            }
        HIP_TRY(e, hipMemcpy(e->D.covT, covT.data(), covT.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->D.fac, LT.data(), LT.size() * sizeof(double), hipMemcpyHostToDevice));
        return LMC_OK;
    }
    if (e->cfg.potential == LMC_POT_FULL_F64) {
        // QuadPotentialFull(cov, dtype="float64"): L = cholesky(cov) in float64 (quadpotential.py:441-443); velocity = cov x
        // (:446-448); random = solve_triangular(L^T, n) (:450-453) = L^-T n, served by the rows of L^-1 (extended precision)
        std::vector<double> L(m);
        if (!host_cholesky(L, d)) return fail(e, LMC_ERR_INVALID, "matrix is not positive definite");
        std::vector<long double> Li(dd, 0.0L);
        for (int c = 0; c < d; ++c)
            for (int i = c; i < d; ++i) {
                long double acc = (i == c) ? 1.0L : 0.0L;
                for (int k = c; k < i; ++k) acc -= static_cast<long double>(L[static_cast<size_t>(i) * d + k]) * Li[static_cast<size_t>(k) * d + c];
                Li[static_cast<size_t>(i) * d + c] = acc / L[static_cast<size_t>(i) * d + i];
            }
        const size_t drows = sweep_rows(d);
        std::vector<double> covT(drows * dp, 0.0), finv(drows * dp, 0.0), LT(drows * dp, 0.0);
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                covT[static_cast<size_t>(j) * dp + i] = m[static_cast<size_t>(i) * d + j];
                LT[static_cast<size_t>(j) * dp + i] = (j <= i) ? L[static_cast<size_t>(i) * d + j] : 0.0;
                finv[static_cast<size_t>(i) * dp + j] = (j <= i) ? static_cast<double>(Li[static_cast<size_t>(i) * d + j]) : 0.0;   // row k of L^-1 over i
            }
        HIP_TRY(e, hipMemcpy(e->D.covT, covT.data(), covT.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->D.fac, finv.data(), finv.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->chol64T, LT.data(), LT.size() * sizeof(double), hipMemcpyHostToDevice));
        return LMC_OK;
    }
    if (e->cfg.potential == LMC_POT_FULL_ADAPT && e->D.mat_f64) {
        // QuadPotentialFullAdapt(dtype="float64") (quadpotential.py:507-509): float64 covariance and factor
        if (adaptation_window < 1 || update_window < 1 || !(adaptation_window_multiplier > 0.0) || initial_weight < 0.0)
            return fail(e, LMC_ERR_INVALID, "bad FullAdapt parameters");
        std::vector<double> L(m);
        if (!host_cholesky(L, d)) return fail(e, LMC_ERR_INVALID, "matrix is not positive definite");
        std::vector<double> covT(static_cast<size_t>(d) * dp, 0.0), fac(static_cast<size_t>(d8) * dp, 0.0);
        std::vector<double> rawT(static_cast<size_t>(d) * dp, 0.0), mean_p(dp, 0.0);
        for (int i = 0; i < d; ++i) {
            mean_p[i] = mean[i];
            for (int j = 0; j < d; ++j) {
                covT[static_cast<size_t>(j) * dp + i] = m[static_cast<size_t>(i) * d + j];
                rawT[static_cast<size_t>(j) * dp + i] = m[static_cast<size_t>(i) * d + j];
                fac[static_cast<size_t>(i) * dp + j] = L[static_cast<size_t>(i) * d + j];
            }
        }
        for (int i = d; i < d8; ++i) fac[static_cast<size_t>(i) * dp + i] = 1.0;
        HIP_TRY(e, hipMemcpy(e->cov1T, covT.data(), covT.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->fac1, fac.data(), fac.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->raw1T, rawT.data(), rawT.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->mean1, mean_p.data(), mean_p.size() * sizeof(double), hipMemcpyHostToDevice));
        e->dense_weight = initial_weight;
        e->dense_window = adaptation_window;
        e->dense_multiplier = adaptation_window_multiplier;
        e->dense_update_window = update_window;
        const int rc64 = dense_reset(e);
        if (rc64 != LMC_OK) return rc64;
        HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
        return LMC_OK;
    }
    // QuadPotentialFull / FullAdapt: float32 covariance and its lower Cholesky factor (quadpotential.py:441-443)
    std::vector<float> cov(dd);
    for (size_t i = 0; i < dd; ++i) cov[i] = static_cast<float>(m[i]);
    std::vector<float> L(cov);
    if (!host_cholesky(L, d)) return fail(e, LMC_ERR_INVALID, "matrix is not positive definite");
    std::vector<float> covT(static_cast<size_t>(d) * dp, 0.0f), fac(static_cast<size_t>(d8) * dp, 0.0f);
    std::vector<double> rawT(static_cast<size_t>(d) * dp, 0.0), mean_p(dp, 0.0);
    for (int i = 0; i < d; ++i) {
        mean_p[i] = mean[i];
        for (int j = 0; j < d; ++j) {
            covT[static_cast<size_t>(j) * dp + i] = cov[static_cast<size_t>(i) * d + j];
            rawT[static_cast<size_t>(j) * dp + i] = m[static_cast<size_t>(i) * d + j];      // float64 initial_cov (:506-508)
            fac[static_cast<size_t>(i) * dp + j] = L[static_cast<size_t>(i) * d + j];
        }
    }
    for (int i = d; i < d8; ++i) fac[static_cast<size_t>(i) * dp + i] = 1.0f;   // padding rows: identity
    if (e->cfg.potential == LMC_POT_FULL) {
        HIP_TRY(e, hipMemcpy(e->D.covT, covT.data(), covT.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->D.fac, fac.data(), fac.size() * sizeof(float), hipMemcpyHostToDevice));
        // L^-1 of the float32 factor, extended precision, for the coop kernel's momentum draw (lmc_dense_types.hpp)
        std::vector<long double> Li(dd, 0.0L);
        for (int c = 0; c < d; ++c)
            for (int i = c; i < d; ++i) {
                long double acc = (i == c) ? 1.0L : 0.0L;
                for (int k = c; k < i; ++k) acc -= static_cast<long double>(L[static_cast<size_t>(i) * d + k]) * Li[static_cast<size_t>(k) * d + c];
                Li[static_cast<size_t>(i) * d + c] = acc / static_cast<long double>(L[static_cast<size_t>(i) * d + i]);
            }
        std::vector<double> finv(static_cast<size_t>(sweep_rows(d)) * dp, 0.0);
        for (int k = 0; k < d; ++k)
            for (int i = 0; i <= k; ++i) finv[static_cast<size_t>(k) * dp + i] = static_cast<double>(Li[static_cast<size_t>(k) * d + i]);
        HIP_TRY(e, hipMemcpy(const_cast<double*>(e->D.fac_inv), finv.data(), finv.size() * sizeof(double), hipMemcpyHostToDevice));
        return LMC_OK;
    }
    if (adaptation_window < 1 || update_window < 1 || !(adaptation_window_multiplier > 0.0) || initial_weight < 0.0)
        return fail(e, LMC_ERR_INVALID, "bad FullAdapt parameters");
    HIP_TRY(e, hipMemcpy(e->cov1T, covT.data(), covT.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->fac1, fac.data(), fac.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->raw1T, rawT.data(), rawT.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->mean1, mean_p.data(), mean_p.size() * sizeof(double), hipMemcpyHostToDevice));
    e->dense_weight = initial_weight;
    e->dense_window = adaptation_window;
    e->dense_multiplier = adaptation_window_multiplier;
    e->dense_update_window = update_window;
    int rc = dense_reset(e);
    if (rc != LMC_OK) return rc;
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

int lmc_engine_dense_update(lmc_engine* e, int32_t tune) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    if (e->cfg.potential < LMC_POT_FULL) return fail(e, LMC_ERR_STATE, "not a dense potential");
    if (!tune || e->cfg.potential != LMC_POT_FULL_ADAPT) return LMC_OK;   // update() returns at once (quadpotential.py:530-531)
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const int rc = dense_launch_adapt(main_stream(e), e->A, e->D, e->dense_multiplier, e->dense_update_window);
    if (rc != 0) return dense_fail(e, rc, "dense update");
    return LMC_OK;
}

static int dense_state_xfer(lmc_engine* e, const lmc_dense_state* st, bool to_user) {
    if (!e || !st) return fail(e, LMC_ERR_INVALID, "null argument");
    if (e->cfg.potential < LMC_POT_FULL) return fail(e, LMC_ERR_STATE, "not a dense potential");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    const size_t C = e->cfg.chains, d = e->cfg.dim, dp = e->dpad, d8 = e->d8;
    const bool adapt = e->cfg.potential == LMC_POT_FULL_ADAPT;
    const bool inv = pot_f64(e->cfg.potential);   // float64 matrices, factor stored transposed
    const DenseArrays& D = e->D;
    const void* fac_src = e->cfg.potential == LMC_POT_FULL_F64 ? static_cast<const void*>(e->chol64T) : D.fac;
    if (!to_user && !adapt) return fail(e, LMC_ERR_STATE, "only FULL_ADAPT has settable dense state");
    if (!adapt && (st->fore_mean || st->fore_raw_cov || st->fore_n || st->back_mean || st->back_raw_cov || st->back_n ||
                   st->window || st->previous_update || st->chol_failures))
        return fail(e, LMC_ERR_STATE, "estimator fields exist for FULL_ADAPT only");
    const size_t P = adapt ? C : 1;
    const bool dev64 = D.mat_f64 != 0;   // the device matrices hold doubles (FullInv, Full float64, FullAdapt float64)
    // matrices: device [P][rows][dp] (transposed for cov) <-> user [C][d][d], float or double on either side
    auto mat_to_user = [&](void* user, bool user64, const void* dev, size_t rows, bool transpose) -> int {
        std::vector<float> host_f(dev64 ? 0 : P * rows * dp);
        std::vector<double> host_d(dev64 ? P * rows * dp : 0);
        if (dev64) HIP_TRY(e, hipMemcpy(host_d.data(), dev, host_d.size() * sizeof(double), hipMemcpyDeviceToHost));
        else HIP_TRY(e, hipMemcpy(host_f.data(), dev, host_f.size() * sizeof(float), hipMemcpyDeviceToHost));
        std::vector<float> out_f(user64 ? 0 : C * d * d);
        std::vector<double> out_d(user64 ? C * d * d : 0);
        for (size_t c = 0; c < C; ++c) {
            const size_t pc = adapt ? c : 0;
            for (size_t i = 0; i < d; ++i)
                for (size_t j = 0; j < d; ++j) {
                    const size_t src = pc * rows * dp + (transpose ? j * dp + i : i * dp + j);
                    const double v = dev64 ? host_d[src] : static_cast<double>(host_f[src]);
                    if (user64) out_d[(c * d + i) * d + j] = v;
                    else out_f[(c * d + i) * d + j] = static_cast<float>(v);
                }
        }
        if (user64) HIP_TRY(e, hipMemcpy(user, out_d.data(), out_d.size() * sizeof(double), hipMemcpyDefault));
        else HIP_TRY(e, hipMemcpy(user, out_f.data(), out_f.size() * sizeof(float), hipMemcpyDefault));
        return LMC_OK;
    };
    int rc;
    std::vector<int> sel(C, 0);
    if (adapt) HIP_TRY(e, hipMemcpy(sel.data(), D.esel, C * sizeof(int), hipMemcpyDeviceToHost));
    if (to_user) {
        const size_t drows = sweep_rows(e->cfg.dim);
        const size_t fac_rows = inv ? drows : d8;
        if (st->cov && (rc = mat_to_user(st->cov, false, D.covT, drows, true)) != LMC_OK) return rc;
        if (st->chol && (rc = mat_to_user(st->chol, false, fac_src, fac_rows, inv)) != LMC_OK) return rc;
        if (st->cov64 && (rc = mat_to_user(st->cov64, true, D.covT, drows, true)) != LMC_OK) return rc;
        if (st->chol64 && (rc = mat_to_user(st->chol64, true, fac_src, fac_rows, inv)) != LMC_OK) return rc;
    } else {
        auto mat_from_user = [&](const void* user, bool user64, void* dev, size_t rows, bool transpose, bool identity_pad) -> int {
            std::vector<float> in_f(user64 ? 0 : C * d * d);
            std::vector<double> in_d(user64 ? C * d * d : 0);
            if (user64) HIP_TRY(e, hipMemcpy(in_d.data(), user, in_d.size() * sizeof(double), hipMemcpyDefault));
            else HIP_TRY(e, hipMemcpy(in_f.data(), user, in_f.size() * sizeof(float), hipMemcpyDefault));
            std::vector<float> host_f(dev64 ? 0 : C * rows * dp, 0.0f);
            std::vector<double> host_d(dev64 ? C * rows * dp : 0, 0.0);
            for (size_t c = 0; c < C; ++c) {
                for (size_t i = 0; i < d; ++i)
                    for (size_t j = 0; j < d; ++j) {
                        const double v = user64 ? in_d[(c * d + i) * d + j] : static_cast<double>(in_f[(c * d + i) * d + j]);
                        const size_t dst = c * rows * dp + (transpose ? j * dp + i : i * dp + j);
                        if (dev64) host_d[dst] = v; else host_f[dst] = static_cast<float>(v);
                    }
                for (size_t i = d; identity_pad && i < rows; ++i) {
                    if (dev64) host_d[c * rows * dp + i * dp + i] = 1.0; else host_f[c * rows * dp + i * dp + i] = 1.0f;
                }
            }
            if (dev64) HIP_TRY(e, hipMemcpy(dev, host_d.data(), host_d.size() * sizeof(double), hipMemcpyHostToDevice));
            else HIP_TRY(e, hipMemcpy(dev, host_f.data(), host_f.size() * sizeof(float), hipMemcpyHostToDevice));
            return LMC_OK;
        };
        if ((st->cov && st->cov64) || (st->chol && st->chol64))
            return fail(e, LMC_ERR_INVALID, "set either the float32 or the float64 form of a matrix");
        if (st->cov && (rc = mat_from_user(st->cov, false, D.covT, sweep_rows(e->cfg.dim), true, false)) != LMC_OK) return rc;
        if (st->chol && (rc = mat_from_user(st->chol, false, D.fac, d8, false, true)) != LMC_OK) return rc;
        if (st->cov64 && (rc = mat_from_user(st->cov64, true, D.covT, sweep_rows(e->cfg.dim), true, false)) != LMC_OK) return rc;
        if (st->chol64 && (rc = mat_from_user(st->chol64, true, D.fac, d8, false, true)) != LMC_OK) return rc;
    }
    if (!adapt) return LMC_OK;
    const size_t mplane = C * d * dp, plane = C * dp;
    // estimators: slot sel[c] is the foreground; set() writes the canonical layout (slot 0 = foreground)
    if (!to_user && (st->fore_mean || st->fore_raw_cov || st->fore_n || st->back_mean || st->back_raw_cov || st->back_n)) {
        if (!(st->fore_mean && st->fore_raw_cov && st->fore_n && st->back_mean && st->back_raw_cov && st->back_n))
            return fail(e, LMC_ERR_INVALID, "the estimator fields must be set together");
        std::fill(sel.begin(), sel.end(), 0);
        HIP_TRY(e, hipMemcpy(D.esel, sel.data(), C * sizeof(int), hipMemcpyHostToDevice));
    }
    auto raw_xfer = [&](double* user, int which) -> int {   // which: 0 foreground, 1 background
        if (!user) return LMC_OK;
        std::vector<double> host(2 * mplane), buf(C * d * d);
        if (to_user) {
            HIP_TRY(e, hipMemcpy(host.data(), D.rawT, host.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t c = 0; c < C; ++c) {
                const size_t slot = which == 0 ? sel[c] : 1 - sel[c];
                for (size_t i = 0; i < d; ++i)
                    for (size_t j = 0; j < d; ++j) buf[(c * d + i) * d + j] = host[slot * mplane + c * d * dp + j * dp + i];
            }
            HIP_TRY(e, hipMemcpy(user, buf.data(), buf.size() * sizeof(double), hipMemcpyDefault));
        } else {
            HIP_TRY(e, hipMemcpy(buf.data(), user, buf.size() * sizeof(double), hipMemcpyDefault));
            std::vector<double> one(mplane, 0.0);
            for (size_t c = 0; c < C; ++c)
                for (size_t i = 0; i < d; ++i)
                    for (size_t j = 0; j < d; ++j) one[c * d * dp + j * dp + i] = buf[(c * d + i) * d + j];
            HIP_TRY(e, hipMemcpy(D.rawT + which * mplane, one.data(), mplane * sizeof(double), hipMemcpyHostToDevice));
        }
        return LMC_OK;
    };
    auto mean_xfer = [&](double* user, int which) -> int {
        if (!user) return LMC_OK;
        std::vector<double> host(2 * plane), buf(C * d);
        if (to_user) {
            HIP_TRY(e, hipMemcpy(host.data(), D.emean, host.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t c = 0; c < C; ++c) {
                const size_t slot = which == 0 ? sel[c] : 1 - sel[c];
                for (size_t i = 0; i < d; ++i) buf[c * d + i] = host[slot * plane + c * dp + i];
            }
            HIP_TRY(e, hipMemcpy(user, buf.data(), buf.size() * sizeof(double), hipMemcpyDefault));
        } else {
            HIP_TRY(e, hipMemcpy(buf.data(), user, buf.size() * sizeof(double), hipMemcpyDefault));
            std::vector<double> one(plane, 0.0);
            for (size_t c = 0; c < C; ++c)
                for (size_t i = 0; i < d; ++i) one[c * dp + i] = buf[c * d + i];
            HIP_TRY(e, hipMemcpy(D.emean + which * plane, one.data(), plane * sizeof(double), hipMemcpyHostToDevice));
        }
        return LMC_OK;
    };
    auto n_xfer = [&](double* user, int which) -> int {
        if (!user) return LMC_OK;
        std::vector<double> host(2 * C), buf(C);
        HIP_TRY(e, hipMemcpy(host.data(), D.en, host.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (to_user) {
            for (size_t c = 0; c < C; ++c) buf[c] = host[2 * c + (which == 0 ? sel[c] : 1 - sel[c])];
            HIP_TRY(e, hipMemcpy(user, buf.data(), C * sizeof(double), hipMemcpyDefault));
        } else {
            HIP_TRY(e, hipMemcpy(buf.data(), user, C * sizeof(double), hipMemcpyDefault));
            for (size_t c = 0; c < C; ++c) host[2 * c + which] = buf[c];
            HIP_TRY(e, hipMemcpy(D.en, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        return LMC_OK;
    };
    if ((rc = raw_xfer(st->fore_raw_cov, 0)) != LMC_OK) return rc;
    if ((rc = raw_xfer(st->back_raw_cov, 1)) != LMC_OK) return rc;
    if ((rc = mean_xfer(st->fore_mean, 0)) != LMC_OK) return rc;
    if ((rc = mean_xfer(st->back_mean, 1)) != LMC_OK) return rc;
    if ((rc = n_xfer(st->fore_n, 0)) != LMC_OK) return rc;
    if ((rc = n_xfer(st->back_n, 1)) != LMC_OK) return rc;
    auto ints = [&](int32_t* user, int* dev) -> int {
        if (!user) return LMC_OK;
        if (to_user) HIP_TRY(e, hipMemcpy(user, dev, C * sizeof(int), hipMemcpyDefault));
        else HIP_TRY(e, hipMemcpy(dev, user, C * sizeof(int), hipMemcpyDefault));
        return LMC_OK;
    };
    if ((rc = ints(st->window, D.window)) != LMC_OK) return rc;
    if ((rc = ints(st->previous_update, D.prev_update)) != LMC_OK) return rc;
    if ((rc = ints(st->chol_failures, D.chol_failed)) != LMC_OK) return rc;
    return LMC_OK;
}

static int dense_chain_fetch(lmc_engine* e, int32_t chain, void* cov, void* chol, bool user64) {
    if (!e || chain < 0 || chain >= e->cfg.chains) return fail(e, LMC_ERR_INVALID, "bad chain index");
    if (e->cfg.potential < LMC_POT_FULL) return fail(e, LMC_ERR_STATE, "not a dense potential");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    const size_t d = e->cfg.dim, dp = e->dpad;
    const bool inv = pot_f64(e->cfg.potential), dev64 = e->D.mat_f64 != 0;
    const void* fac_src = e->cfg.potential == LMC_POT_FULL_F64 ? static_cast<const void*>(e->chol64T) : e->D.fac;
    const size_t esz = dev64 ? sizeof(double) : sizeof(float);
    auto fetch = [&](void* user, const void* dev, long long stride, bool transpose) -> int {
        std::vector<char> raw(d * dp * esz);
        const char* src = static_cast<const char*>(dev) + static_cast<size_t>(chain) * static_cast<size_t>(stride) * esz;
        HIP_TRY(e, hipMemcpy(raw.data(), src, raw.size(), hipMemcpyDeviceToHost));
        std::vector<float> out_f(user64 ? 0 : d * d);
        std::vector<double> out_d(user64 ? d * d : 0);
        for (size_t i = 0; i < d; ++i)
            for (size_t j = 0; j < d; ++j) {
                const size_t k = transpose ? j * dp + i : i * dp + j;
                const double v = dev64 ? reinterpret_cast<const double*>(raw.data())[k]
                                       : static_cast<double>(reinterpret_cast<const float*>(raw.data())[k]);
                if (user64) out_d[i * d + j] = v; else out_f[i * d + j] = static_cast<float>(v);
            }
        if (user64) HIP_TRY(e, hipMemcpy(user, out_d.data(), out_d.size() * sizeof(double), hipMemcpyDefault));
        else HIP_TRY(e, hipMemcpy(user, out_f.data(), out_f.size() * sizeof(float), hipMemcpyDefault));
        return LMC_OK;
    };
    int rc;
    if (cov && (rc = fetch(cov, e->D.covT, e->D.mat_stride, true)) != LMC_OK) return rc;
    if (chol && (rc = fetch(chol, fac_src, e->D.fac_stride, inv)) != LMC_OK) return rc;
    return LMC_OK;
}

int lmc_engine_get_dense_chain(lmc_engine* e, int32_t chain, float* cov, float* chol) {
    return dense_chain_fetch(e, chain, cov, chol, false);
}
int lmc_engine_get_dense_chain_f64(lmc_engine* e, int32_t chain, double* cov, double* chol) {
    return dense_chain_fetch(e, chain, cov, chol, true);
}

int lmc_engine_get_dense_factor_f64(lmc_engine* e, double* chol) {
    if (!e || !chol) return fail(e, LMC_ERR_INVALID, "null argument");
    if (!pot_f64(e->cfg.potential)) return fail(e, LMC_ERR_STATE, "the float64 factor exists for LMC_POT_FULL_F64 / LMC_POT_FULL_INV");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    const size_t d = e->cfg.dim, dp = e->dpad;
    const double* src = e->cfg.potential == LMC_POT_FULL_F64 ? e->chol64T : static_cast<const double*>(e->D.fac);   // LT[j][i] = L[i][j]
    std::vector<double> raw(d * dp), out(d * d);
    HIP_TRY(e, hipMemcpy(raw.data(), src, raw.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < d; ++j) out[i * d + j] = raw[j * dp + i];
    HIP_TRY(e, hipMemcpy(chol, out.data(), out.size() * sizeof(double), hipMemcpyDefault));
    return LMC_OK;
}

int lmc_engine_get_dense_state(lmc_engine* e, const lmc_dense_state* dst) { return dense_state_xfer(e, dst, true); }
int lmc_engine_set_dense_state(lmc_engine* e, const lmc_dense_state* src) { return dense_state_xfer(e, src, false); }

int lmc_engine_seed(lmc_engine* e, const uint32_t* seeds) {
    if (!e || !seeds) return fail(e, LMC_ERR_INVALID, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    uint32_t* dseeds = nullptr;
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&dseeds), e->cfg.chains * sizeof(uint32_t)));
    hipError_t err = hipMemcpyAsync(dseeds, seeds, e->cfg.chains * sizeof(uint32_t), hipMemcpyDefault, main_stream(e));
    if (err == hipSuccess) {
        LMC_LAUNCH(seed_kernel, dim3(e->cfg.chains), dim3(64), 0, main_stream(e), e->A, dseeds);
        err = hipGetLastError();
        if (err == hipSuccess)
            err = hipMemcpyAsync(e->seeds, dseeds, e->cfg.chains * sizeof(uint32_t), hipMemcpyDeviceToDevice, main_stream(e));
    }
    if (err == hipSuccess) err = hipStreamSynchronize(main_stream(e));
    (void)hipFree(dseeds);
    if (err != hipSuccess) return fail(e, LMC_ERR_HIP, "seed: %s", hipGetErrorString(err));
    // new seeds = a new job: what the chains of the previous one reported about their tree sizes says nothing about it
    if (e->stop_host) __atomic_store_n(e->stop_host + 24, 0, __ATOMIC_RELEASE);
    if (e->lds_plan == 2) e->plan_now = 0;
    return LMC_OK;
}

int lmc_engine_set_rng_state(lmc_engine* e, int32_t chain, const uint32_t* key, int32_t pos, int32_t has_gauss,
                             double gauss) {
    if (!e || !key || chain < 0 || chain >= e->cfg.chains || pos < 0 || pos > kMtN)
        return fail(e, LMC_ERR_INVALID, "bad rng state");
    if ((pos & 1) && e->run_w > 1)   // the team kernels twist between barriers at even positions only
        return fail(e, LMC_ERR_INVALID, "odd MT19937 position %d (a 32-bit legacy draw, e.g. np.random.randint, came before): "
                    "chains of more than 256 dimensions need an even position -- draw one more 32-bit value first", pos);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    HIP_TRY(e, hipMemcpy(e->A.mt + static_cast<size_t>(chain) * kMtN, key, kMtN * sizeof(uint32_t), hipMemcpyDefault));
    HIP_TRY(e, hipMemcpy(e->A.rng_pos + chain, &pos, sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->A.rng_has_gauss + chain, &has_gauss, sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->A.rng_gauss + chain, &gauss, sizeof(double), hipMemcpyHostToDevice));
    return LMC_OK;
}

int lmc_engine_get_rng_state(lmc_engine* e, int32_t chain, uint32_t* key, int32_t* pos, int32_t* has_gauss,
                             double* gauss) {
    if (!e || chain < 0 || chain >= e->cfg.chains) return fail(e, LMC_ERR_INVALID, "bad chain index");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    if (key) HIP_TRY(e, hipMemcpy(key, e->A.mt + static_cast<size_t>(chain) * kMtN, kMtN * sizeof(uint32_t), hipMemcpyDefault));
    if (pos) HIP_TRY(e, hipMemcpy(pos, e->A.rng_pos + chain, sizeof(int), hipMemcpyDeviceToHost));
    if (has_gauss) HIP_TRY(e, hipMemcpy(has_gauss, e->A.rng_has_gauss + chain, sizeof(int), hipMemcpyDeviceToHost));
    if (gauss) HIP_TRY(e, hipMemcpy(gauss, e->A.rng_gauss + chain, sizeof(double), hipMemcpyDeviceToHost));
    return LMC_OK;
}

int lmc_engine_set_position(lmc_engine* e, const double* q, int32_t per_chain) {
    if (!e || !q) return fail(e, LMC_ERR_INVALID, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const size_t C = e->cfg.chains, d = e->cfg.dim, dp = e->dpad;
    HIP_TRY(e, hipMemsetAsync(e->A.q, 0, C * dp * sizeof(double), main_stream(e)));
    HIP_TRY(e, hipMemsetAsync(e->A.status, 0, C * sizeof(int), main_stream(e)));   // new positions: per-chain failure bits start clean
    if (per_chain) {
        HIP_TRY(e, hipMemcpy2DAsync(e->A.q, dp * sizeof(double), q, d * sizeof(double), d * sizeof(double), C,
                                    hipMemcpyDefault, main_stream(e)));
    } else {   // same start for every chain (sampling.py:163-164): pitch 0 is not portable, loop rows
        std::vector<double> host(d);
        HIP_TRY(e, hipMemcpy(host.data(), q, d * sizeof(double), hipMemcpyDefault));
        std::vector<double> all(C * d);
        for (size_t c = 0; c < C; ++c) std::memcpy(all.data() + c * d, host.data(), d * sizeof(double));
        HIP_TRY(e, hipMemcpy2DAsync(e->A.q, dp * sizeof(double), all.data(), d * sizeof(double), d * sizeof(double), C,
                                    hipMemcpyHostToDevice, main_stream(e)));
        HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    }
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

int lmc_engine_get_position(lmc_engine* e, double* q) {
    if (!e || !q) return fail(e, LMC_ERR_INVALID, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const size_t C = e->cfg.chains, d = e->cfg.dim, dp = e->dpad;
    HIP_TRY(e, hipMemcpy2DAsync(q, d * sizeof(double), e->A.q, dp * sizeof(double), d * sizeof(double), C,
                                hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

int lmc_engine_reset_tuning(lmc_engine* e) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (e->stop_host) __atomic_store_n(e->stop_host + 16, 0, __ATOMIC_RELEASE);   // progress hint: iteration 0 again
    if (e->stop_host) __atomic_store_n(e->stop_host + 24, 0, __ATOMIC_RELEASE);   // tree-size hint: nothing reported yet
    if (e->lds_plan == 2) e->plan_now = 0;
    e->plan_last = -1;
    // QuadPotentialDiag.reset() is a no-op (quadpotential.py:138-140): only the adaptive potential resets
    int rc = launch_reset(e, 1, e->cfg.potential == LMC_POT_DIAG_ADAPT ? 1 : 0);
    if (rc != LMC_OK) return rc;
    // FullAdapt: the reference's reset() is the inherited no-op, so its sequential driver hands chain k the matrix
    // of chain k-1; chains here are independent, each one a fresh one-chain run (what the reference's
    // multi-process driver computes): restore the constructor state
    if (e->cfg.potential == LMC_POT_FULL_ADAPT) return dense_reset(e);
    return LMC_OK;
}

int lmc_engine_set_dual_average(lmc_engine* e, double log_step, double log_bar, double hbar, int32_t count) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const int threads = 256, blocks = (e->cfg.chains + threads - 1) / threads;
    LMC_LAUNCH(set_da_kernel, dim3(blocks), dim3(threads), 0, main_stream(e), e->A, log_step, log_bar, hbar, count);
    HIP_TRY(e, hipGetLastError());
    return LMC_OK;
}

int lmc_engine_reserve(lmc_engine* e, int64_t capacity, int64_t trace_begin) {
    if (!e || capacity < 1) return fail(e, LMC_ERR_INVALID, "capacity must be >= 1");
    const bool keep_trace = trace_begin >= 0 && trace_begin < capacity;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    ChainArrays& A = e->A;
    if (!e->trace_external) dev_free(e, A.trace);
    A.trace = nullptr;
    e->trace_external = false;
    dev_free(e, A.stat_rec); A.stat_rec = nullptr;
    const size_t C = e->cfg.chains, cap = static_cast<size_t>(capacity);
    int rc;
    if (keep_trace && (rc = dev_alloc(e, &A.trace, C * (cap - static_cast<size_t>(trace_begin)) * e->cfg.dim, false)) != LMC_OK) return rc;
    A.trace_begin = keep_trace ? trace_begin : 0;
    if ((rc = dev_alloc(e, &A.stat_rec, C * cap)) != LMC_OK) return rc;
    A.cap = capacity;
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

// device_view_of(): below (streamed results)
static void* device_view_of(void* p);

int lmc_engine_attach_trace(lmc_engine* e, double* dst, int64_t trace_begin) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    if (e->A.cap <= 0 || !e->A.stat_rec) return fail(e, LMC_ERR_STATE, "lmc_engine_reserve() must be called before attach_trace()");
    if (trace_begin < 0 || trace_begin >= e->A.cap)
        return fail(e, LMC_ERR_INVALID, "trace_begin %lld outside [0, %lld)", (long long)trace_begin, (long long)e->A.cap);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    ChainArrays& A = e->A;
    double* view = nullptr;
    if (dst) {   // (checked before anything is given up: a refused destination leaves the engine as it was)
        view = static_cast<double*>(device_view_of(dst));
        if (!view) return fail(e, LMC_ERR_INVALID, "trace: the destination is not device-accessible memory (use lmc_host_alloc / lmc_host_register)");
    }
    if (A.trace && !e->trace_external) {   // an engine-owned trace may still be written by launches in flight
        HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
        dev_free(e, A.trace);
    }
    A.trace = nullptr;
    e->trace_external = false;
    if (dst) {
        A.trace = view;
        e->trace_external = true;
    } else {
        const size_t C = e->cfg.chains;
        const int rc = dev_alloc(e, &A.trace, C * static_cast<size_t>(A.cap - trace_begin) * e->cfg.dim, false);
        if (rc != LMC_OK) return rc;
    }
    A.trace_begin = trace_begin;
    return LMC_OK;
}

static SamplerParams make_params(const lmc_engine* e, int64_t n_tune, int64_t iter_begin, int32_t n_iters) {
    SamplerParams P;
    std::memset(&P, 0, sizeof(P));
    P.kind = e->cfg.kind;
    P.momentum_f32 = e->cfg.potential == LMC_POT_DIAG_ADAPT && !e->cfg.mass_f64;   // quadpotential.py:223: normals.astype(dtype)
    P.adapt_mass = e->cfg.potential == LMC_POT_DIAG_ADAPT;
    P.mass_f64 = e->cfg.mass_f64 ? 1 : 0;
    P.adapt_step_size = e->cfg.adapt_step_size;
    P.target_accept = e->cfg.target_accept;
    P.emax = e->cfg.emax;
    P.gamma = e->cfg.gamma;
    P.k = e->cfg.k;
    P.t0 = e->cfg.t0;
    P.max_treedepth = e->cfg.max_treedepth;
    P.early_max_treedepth = e->cfg.early_max_treedepth;
    P.path_length = e->cfg.path_length;
    P.max_steps = e->cfg.max_steps;
    P.window = e->cfg.adaptation_window;
    P.window_multiplier = e->cfg.adaptation_window_multiplier;
    P.n_tune = n_tune;
    P.iter_begin = iter_begin;
    P.n_iters = n_iters;
    P.nlds = e->nlds;
    P.lds_doubles = e->lds_bytes / 8;
    P.sdot_mode = e->cfg.start_energy_sdot;
    P.rng_mode = e->cfg.rng_mode;
    P.step_jitter = e->step_jitter;
    P.jitter_lo = e->jitter_lo;
    P.jitter_hi = e->jitter_hi;
    P.relay_mask = 255;
    return P;
}

// lmc_engine_run() for the shapes the fused kernels are not instantiated for (lmc_wide.hpp)
static int wide_run(lmc_engine* e, SamplerParams P) {
    if (e->cfg.potential >= LMC_POT_FULL) P.momentum_f32 = !e->D.mat_f64;
    if (e->cfg.potential != LMC_POT_DIAG_ADAPT) P.adapt_mass = 0;
    P.chain_begin = 0;
    P.relay_mask = relay_mask_for(e->cfg.chains);
    hipStream_t st = main_stream(e);
    const bool rtc = e->cfg.target_family == LMC_TARGET_USER && !kUserCompiledInDense;
    // while tuning, FullAdapt refreshes covariance and factor after EVERY iteration (quadpotential.py:528-552): one iteration
    // per launch with the update kernel in between, like dense_run(); everything else is one launch
    const long long end = P.iter_begin + P.n_iters;
    long long it = P.iter_begin;
    while (it < end) {
        const bool adapt = e->cfg.potential == LMC_POT_FULL_ADAPT && it < P.n_tune;
        SamplerParams Q = P;
        Q.iter_begin = it;
        Q.n_iters = static_cast<int>(adapt ? 1 : end - it);
        int rc;
        if (rtc) {
            void* args[] = {&e->A, &e->D, &Q, &e->tparams};
            rc = user_launch(e, e->user_run, st, static_cast<unsigned>(e->cfg.chains), 64u * e->run_w, static_cast<unsigned>(e->lds_bytes), args);
            if (rc != LMC_OK) return rc;
        } else {
            rc = wide_launch_run(e->cfg.target_family, e->ns, e->run_w, st, e->A, e->D, Q, e->tparams, e->cfg.chains);
            if (rc != 0) return dense_fail(e, rc, "run (general kernel)");
        }
        if (adapt) {
            rc = dense_launch_adapt(st, e->A, e->D, e->dense_multiplier, e->dense_update_window, nullptr, 0, e->cfg.chains,
                                    static_cast<int>(it + 1));
            if (rc != 0) return dense_fail(e, rc, "dense update");
        }
        it += Q.n_iters;
    }
    return LMC_OK;
}

int lmc_engine_run(lmc_engine* e, int64_t n_tune, int64_t iter_begin, int32_t n_iters) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    if (e->A.cap <= 0 || !e->A.stat_rec) return fail(e, LMC_ERR_STATE, "lmc_engine_reserve() must be called before run()");
    if (iter_begin < 0 || n_iters < 0 || iter_begin + n_iters > e->A.cap)
        return fail(e, LMC_ERR_INVALID, "iterations [%lld, %lld) exceed reserved capacity %lld", (long long)iter_begin,
                    (long long)(iter_begin + n_iters), (long long)e->A.cap);
    if (n_iters == 0) return LMC_OK;
    if (e->cfg.target_family == LMC_TARGET_EXTERNAL)
        return fail(e, LMC_ERR_STATE, "the density is evaluated by the caller: drive the chains with lmc_engine_tick_begin() / lmc_engine_tick()");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    SamplerParams P = make_params(e, n_tune, iter_begin, n_iters);
    if (e->wide) return wide_run(e, P);
    if (e->cfg.potential >= LMC_POT_FULL) return dense_run(e, P);
    int plan = choose_lds_plan(e, iter_begin);
    // the deep-tree plan exists for the one-wave kernels, and for a run-time compiled density only once its plan-1 kernel has
    // been handed over: anything else launches under plan 0 whatever was chosen (LDS layout and kernel must agree)
    if (e->run_w != 1 || e->lds_bytes1 <= 0 ||
        (e->cfg.target_family == LMC_TARGET_USER && !kUserCompiledIn && !e->user_run1))
        plan = 0;
    e->plan_last = plan;
    if (plan == 1) {   // the deep-tree plan: its own level count, no generator behind the stack
        P.nlds = e->nlds1;
        P.lds_doubles = e->lds_bytes1 / 8;
    }
    const int run_lds = sampling_lds_bytes(e, plan);   // subtree stack (+ MT19937 + team exchange under plan 0)
    const dim3 block(64 * e->run_w);
    const int n_sub = e->n_sub;
    HIP_TRY(e, order_sub_blocks_after_main(e));
#define RUN_PLAN1(NSV, WV, T)                                                                                  \
    if constexpr (WV == 1) {                                                                                   \
        LMC_LAUNCH((run_kernel<NSV, 1, T, 0, 1>), grid, block, run_lds, st, e->A, P, e->tparams);              \
    }
#define RUN_ONE(NSV, WV, T)                                                                                    \
    if (!found && e->run_ns == NSV && e->run_w == WV) {                                                        \
        found = true;                                                                                          \
        if (e->cfg.rng_mode == LMC_RNG_PHILOX) {                                                               \
            if (run_lds > 64 * 1024)                                                                           \
                HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&run_kernel<NSV, WV, T, 1>),      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, run_lds));          \
            LMC_LAUNCH((run_kernel<NSV, WV, T, 1>), grid, block, run_lds, st, e->A, P, e->tparams);            \
        } else if (plan == 1) {                                                                                \
            RUN_PLAN1(NSV, WV, T)                                                                              \
        } else {                                                                                               \
            if (run_lds > 64 * 1024)                                                                           \
                HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&run_kernel<NSV, WV, T>),         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, run_lds));          \
            LMC_LAUNCH((run_kernel<NSV, WV, T>), grid, block, run_lds, st, e->A, P, e->tparams);               \
        }                                                                                                      \
    }
#define RUN_CALL(T)                                                                                            \
    {                                                                                                          \
        bool found = false;                                                                                    \
        LMC_FOR_EACH_SHAPE(RUN_ONE, T)                                                                         \
        if (!found) return fail(e, LMC_ERR_INVALID, "unsupported kernel shape ns=%d w=%d", e->run_ns, e->run_w); \
    }
    for (int b = 0; b < n_sub; ++b) {
        const long long lo = static_cast<long long>(e->cfg.chains) * b / n_sub, hi = static_cast<long long>(e->cfg.chains) * (b + 1) / n_sub;
        P.chain_begin = static_cast<int>(lo);
        P.relay_mask = relay_mask_for(hi - lo);
        const dim3 grid(static_cast<unsigned>(hi - lo));
        hipStream_t st = n_sub > 1 ? e->sub_stream[b] : main_stream(e);
        if (n_sub > 1) e->sub_pending = true;
        if (e->cfg.target_family == LMC_TARGET_USER && !kUserCompiledIn) {
            void* args[] = {&e->A, &P, &e->tparams};
            const int rc = user_launch(e, plan == 1 ? e->user_run1 : e->user_run, st, grid.x, block.x, static_cast<unsigned>(run_lds), args);
            if (rc != LMC_OK) return rc;
            continue;
        }
        LMC_FAMILY_SWITCH(e, e->cfg.target_family, RUN_CALL)
    }
#undef RUN_CALL
#undef RUN_ONE
#undef RUN_PLAN1
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, order_external_stream_after_sub_blocks(e));
    return LMC_OK;
}

// The streams run() launches on (one per sub-block; the main stream if the engine does not split its chains): a caller
// that times launches records its events there.
int lmc_engine_run_streams(lmc_engine* e, void** streams, int32_t capacity) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    const int n = e->n_sub > 1 ? e->n_sub : 1;
    if (streams)
        for (int b = 0; b < n && b < capacity; ++b) streams[b] = e->n_sub > 1 ? static_cast<void*>(e->sub_stream[b]) : static_cast<void*>(e->stream_);
    return n;
}

// ---- externally evaluated density: the resumable sampler (lmc_tick.hpp) -----------------------------------
int lmc_engine_tick_begin(lmc_engine* e, int64_t n_tune, int64_t iter_begin, int32_t n_iters) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    if (e->cfg.target_family != LMC_TARGET_EXTERNAL)
        return fail(e, LMC_ERR_STATE, "lmc_engine_tick*() needs cfg.target_family = LMC_TARGET_EXTERNAL");
    if (e->A.cap <= 0 || !e->A.stat_rec) return fail(e, LMC_ERR_STATE, "lmc_engine_reserve() must be called before tick_begin()");
    if (iter_begin < 0 || n_iters < 0 || iter_begin + n_iters > e->A.cap)
        return fail(e, LMC_ERR_INVALID, "iterations [%lld, %lld) exceed reserved capacity %lld", (long long)iter_begin,
                    (long long)(iter_begin + n_iters), (long long)e->A.cap);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    e->K.iter_end = iter_begin + n_iters;
    e->K.n_tune = n_tune;
    const int rc = e->wide ? tick_wide_launch_begin(e->ns, main_stream(e), e->A, e->K, iter_begin)
                           : tick_launch_begin(e->ns, main_stream(e), e->A, e->K, iter_begin);
    if (rc != 0) return fail(e, LMC_ERR_HIP, "tick_begin: %s", rc < 0 ? "unsupported vector width" : hipGetErrorString(static_cast<hipError_t>(rc)));
    e->ticking = true;
    return LMC_OK;
}

void* lmc_engine_tick_positions(lmc_engine* e) { return e ? e->K.q_eval : nullptr; }

int lmc_engine_tick(lmc_engine* e, const double* logp, const double* grad, int32_t* n_active) {
    if (!e || !logp || !grad) return fail(e, LMC_ERR_INVALID, "null argument");
    if (!e->ticking) return fail(e, LMC_ERR_STATE, "lmc_engine_tick_begin() must be called first");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    SamplerParams P = make_params(e, e->K.n_tune, 0, 0);
    if (e->cfg.potential >= LMC_POT_FULL) {
        P.momentum_f32 = !pot_f64(e->cfg.potential);
        P.adapt_mass = 0;
        int rc = tick_dense_launch(e->ns, pot_f64(e->cfg.potential), main_stream(e), e->A, e->D, e->K, P, logp, grad,
                                   e->adapt_mask);
        if (rc != 0) return dense_fail(e, rc, "tick");
        if (e->cfg.potential == LMC_POT_FULL_ADAPT) {   // update() of the chains that finished a tuning iteration in this tick
            rc = dense_launch_adapt(main_stream(e), e->A, e->D, e->dense_multiplier, e->dense_update_window, e->adapt_mask);
            if (rc != 0) return dense_fail(e, rc, "dense update");
        }
    } else {
        const int rc = e->wide ? tick_wide_launch(e->ns, main_stream(e), e->A, e->K, P, logp, grad)
                               : tick_launch(e->ns, main_stream(e), e->A, e->K, P, logp, grad);
        if (rc != 0) return fail(e, LMC_ERR_HIP, "tick: %s", rc < 0 ? "unsupported vector width" : hipGetErrorString(static_cast<hipError_t>(rc)));
    }
    if (n_active) {
        HIP_TRY(e, hipMemsetAsync(e->K.n_active, 0, sizeof(int), main_stream(e)));
        const int rc2 = tick_launch_count(main_stream(e), e->K, e->cfg.chains);
        if (rc2 != 0) return fail(e, LMC_ERR_HIP, "tick count: %s", hipGetErrorString(static_cast<hipError_t>(rc2)));
        HIP_TRY(e, hipMemcpyAsync(n_active, e->K.n_active, sizeof(int), hipMemcpyDeviceToHost, main_stream(e)));
        HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    }
    return LMC_OK;
}

static int copy_rows(lmc_engine* e, void* dst, const void* src, size_t elem, int64_t iter_begin, int64_t n_iters,
                     size_t per_iter) {
    // src layout [C][cap][per_iter], dst layout [C][n_iters][per_iter]
    const size_t C = e->cfg.chains;
    const char* s = static_cast<const char*>(src) + static_cast<size_t>(iter_begin) * per_iter * elem;
    HIP_TRY(e, hipMemcpy2DAsync(dst, n_iters * per_iter * elem, s, e->A.cap * per_iter * elem, n_iters * per_iter * elem,
                                C, hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

static int check_window(lmc_engine* e, const void* dst, int64_t iter_begin, int64_t n_iters) {
    if (!e || !dst) return fail(e, LMC_ERR_INVALID, "null argument");
    if (iter_begin < 0 || n_iters < 1 || iter_begin + n_iters > e->A.cap)
        return fail(e, LMC_ERR_INVALID, "window [%lld, %lld) outside reserved capacity %lld", (long long)iter_begin,
                    (long long)(iter_begin + n_iters), (long long)e->A.cap);
    hipError_t err = hipSetDevice(e->cfg.device);
    if (err != hipSuccess) return fail(e, LMC_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(err));
    return LMC_OK;
}

int lmc_engine_get_trace(lmc_engine* e, double* dst, int64_t iter_begin, int64_t n_iters) {
    int rc = check_window(e, dst, iter_begin, n_iters);
    if (rc != LMC_OK) return rc;
    if (!e->A.trace) return fail(e, LMC_ERR_STATE, "no trace was reserved (trace_begin < 0)");
    if (iter_begin < e->A.trace_begin)
        return fail(e, LMC_ERR_INVALID, "draws before iteration %lld were not stored", (long long)e->A.trace_begin);
    const size_t C = e->cfg.chains, d = e->cfg.dim;
    const size_t rows = static_cast<size_t>(e->A.cap - e->A.trace_begin);
    const double* src = e->A.trace + static_cast<size_t>(iter_begin - e->A.trace_begin) * d;
    HIP_TRY(e, hipMemcpy2DAsync(dst, n_iters * d * sizeof(double), src, rows * d * sizeof(double),
                                n_iters * d * sizeof(double), C, hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

// one statistic out of the per-draw records (lmc_sampler.hpp: StatRecord) into the caller's [chains][n_iters] array
static int gather_stat(lmc_engine* e, void* dst, size_t elem, int kind, int idx, int64_t iter_begin, int64_t n_iters) {
    const size_t n = static_cast<size_t>(e->cfg.chains) * static_cast<size_t>(n_iters);
    DevBuf<char> tmp;
    HIP_TRY(e, tmp.alloc(n * elem));
    hipStream_t st = main_stream(e);
    LMC_LAUNCH(stat_gather_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, e->A.stat_rec, e->A.cap,
               e->cfg.chains, iter_begin, n_iters, kind, idx, e->cfg.kind == LMC_KIND_HMC ? 1 : 0, static_cast<void*>(tmp.p));
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipMemcpyAsync(dst, tmp.p, n * elem, hipMemcpyDefault, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    return LMC_OK;
}

int lmc_engine_get_stat_f64(lmc_engine* e, int32_t stat, double* dst, int64_t iter_begin, int64_t n_iters) {
    int rc = check_window(e, dst, iter_begin, n_iters);
    if (rc != LMC_OK) return rc;
    if (stat < 0 || stat >= kNumStatF64) return fail(e, LMC_ERR_INVALID, "unknown f64 stat %d", stat);
    return gather_stat(e, dst, sizeof(double), 0, stat, iter_begin, n_iters);
}

int lmc_engine_get_stat_i32(lmc_engine* e, int32_t stat, int32_t* dst, int64_t iter_begin, int64_t n_iters) {
    int rc = check_window(e, dst, iter_begin, n_iters);
    if (rc != LMC_OK) return rc;
    if (stat < 0 || stat >= kNumStatI32) return fail(e, LMC_ERR_INVALID, "unknown i32 stat %d", stat);
    return gather_stat(e, dst, sizeof(int32_t), 1, stat, iter_begin, n_iters);
}

int lmc_engine_get_stat_u8(lmc_engine* e, int32_t stat, uint8_t* dst, int64_t iter_begin, int64_t n_iters) {
    int rc = check_window(e, dst, iter_begin, n_iters);
    if (rc != LMC_OK) return rc;
    if (stat < 0 || stat >= kNumStatU8) return fail(e, LMC_ERR_INVALID, "unknown u8 stat %d", stat);
    return gather_stat(e, dst, sizeof(uint8_t), 2, stat, iter_begin, n_iters);
}

// ---- streamed results ------------------------------------------------------------------------------------------------------
#ifndef LMC_WINDOW_COPY_BLOCKS
#define LMC_WINDOW_COPY_BLOCKS 64
#endif
#ifndef LMC_WINDOW_COPY_THREADS
#define LMC_WINDOW_COPY_THREADS 64
#endif
static constexpr int kWindowCopyBlocks = LMC_WINDOW_COPY_BLOCKS;   // workgroups of a window copy (lmc_window_dst.copy_workgroups = 0)
static constexpr int kWindowCopyThreads = LMC_WINDOW_COPY_THREADS; // ONE wavefront each: see lmc_engine_copy_window_async

// the address a kernel of this engine's device writes `p` through: device memory as it is, page-locked host memory through its
// device mapping; nullptr for memory the device cannot reach (pageable)
static void* device_view_of(void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) return p;
    if (attr.type == hipMemoryTypeHost) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return dp;
    }
    return nullptr;
}

int lmc_engine_copy_window_async(lmc_engine* e, const lmc_window_dst* dst, int64_t iter_begin, int64_t n_iters) {
    if (!e || !dst) return fail(e, LMC_ERR_INVALID, "null argument");
    if (n_iters == 0) return LMC_OK;
    if (iter_begin < 0 || n_iters < 0 || iter_begin + n_iters > e->A.cap || !e->A.stat_rec)
        return fail(e, LMC_ERR_INVALID, "window [%lld, %lld) outside reserved capacity %lld", (long long)iter_begin,
                    (long long)(iter_begin + n_iters), (long long)e->A.cap);
    if (iter_begin < dst->first || iter_begin + n_iters > dst->first + dst->n_out)
        return fail(e, LMC_ERR_INVALID, "window [%lld, %lld) outside the destination's iterations [%lld, %lld)", (long long)iter_begin,
                    (long long)(iter_begin + n_iters), (long long)dst->first, (long long)(dst->first + dst->n_out));
    if (dst->n_planes < 0 || dst->n_planes > LMC_MAX_PLANES) return fail(e, LMC_ERR_INVALID, "n_planes %d", dst->n_planes);
    if (dst->copy_workgroups < 0 || dst->copy_workgroups > 4096) return fail(e, LMC_ERR_INVALID, "copy_workgroups %d", dst->copy_workgroups);
    if (dst->trace && (!e->A.trace || iter_begin < e->A.trace_begin))
        return fail(e, LMC_ERR_INVALID, "draws before iteration %lld were not stored", (long long)e->A.trace_begin);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    WindowPlanes W;
    std::memset(&W, 0, sizeof(W));
    W.n_planes = dst->n_planes;
    for (int p = 0; p < dst->n_planes; ++p) {
        const lmc_window_plane& pl = dst->plane[p];
        const int lim = pl.kind == LMC_PLANE_F64 ? kNumStatF64 : pl.kind == LMC_PLANE_I32 ? kNumStatI32 : pl.kind == LMC_PLANE_U8 ? kNumStatU8 : -1;
        if (!pl.dst || lim < 0 || pl.idx < 0 || pl.idx >= lim || pl.as < LMC_AS_NATIVE || pl.as > LMC_AS_I64 ||
            (pl.kind != LMC_PLANE_I32 && pl.as != LMC_AS_NATIVE))
            return fail(e, LMC_ERR_INVALID, "plane %d: kind %d idx %d as %d", p, pl.kind, pl.idx, pl.as);
        W.kind[p] = pl.kind; W.idx[p] = pl.idx; W.as[p] = pl.as;
        W.dst[p] = device_view_of(pl.dst);
        if (!W.dst[p])
            return fail(e, LMC_ERR_INVALID, "plane %d: the destination is not device-accessible memory (use lmc_host_alloc)", p);
    }
    double* trace_dst = nullptr;
    if (dst->trace) {
        trace_dst = static_cast<double*>(device_view_of(dst->trace));
        if (!trace_dst) return fail(e, LMC_ERR_INVALID, "trace: the destination is not device-accessible memory (use lmc_host_alloc)");
    }
    // The copies run on a stream of the engine's own, created with HIGH priority: the runtime maps streams of one priority
    // onto a pool of four hardware queues, which the main stream and the four sub-block streams already share -- a copy that
    // waits (for ALL sub-blocks of its launch) in a queue it shares with a sub-block stream holds that sub-block's next launch
    // back, and a copy enqueued on the sub-block streams themselves runs between their launches instead of under them (both
    // measured: the whole copy time showed up in the job). A high-priority stream gets a hardware queue from another pool.
    // Ordered after every launch enqueued so far by one event per sub-block stream (and the main stream), waited for on the
    // device: the host does not block and later launches are not held up.
    if (!e->copy_stream) {
        int pr_low = 0, pr_high = 0;
        HIP_TRY(e, hipDeviceGetStreamPriorityRange(&pr_low, &pr_high));
        HIP_TRY(e, hipStreamCreateWithPriority(&e->copy_stream, hipStreamNonBlocking, pr_high));
        for (hipEvent_t& ev : e->copy_dep) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    hipStream_t cs = e->copy_stream;
    if (e->n_sub > 1)
        for (int b = 0; b < e->n_sub; ++b) {
            HIP_TRY(e, hipEventRecord(e->copy_dep[b], e->sub_stream[b]));
            HIP_TRY(e, hipStreamWaitEvent(cs, e->copy_dep[b], 0));
        }
    HIP_TRY(e, hipEventRecord(e->copy_dep[lmc_engine::kMaxSub], e->stream_));
    HIP_TRY(e, hipStreamWaitEvent(cs, e->copy_dep[lmc_engine::kMaxSub], 0));
    const long long C = e->cfg.chains, d = e->cfg.dim, n = n_iters, n_out = dst->n_out;
    const long long row0 = iter_begin - dst->first;
    // Few workgroups of ONE wavefront each. The copy's wavefronts sit on stores that drain at host-link speed, so a few dozen
    // saturate the link (32 single-wave workgroups reach 40 GiB/s, 64 reach 50) and every further one only takes wave slots
    // from the sampling launches the copy runs under; and a single wavefront finds a slot whenever ONE sampling wavefront
    // retires, where a 1024-thread workgroup needs sixteen free slots on one compute unit -- under a running launch that is
    // the launch's tail (rocprofv3 showed a 115 ms dispatch for an 11 ms copy). Measured under a running job
    // (tools/stream_probe.py, profiles/r06_sample_e2e.txt; 0.34 s of copies of which 0.12 s cannot overlap by construction):
    // 64 x 64 threads +0.18 s, 16 x 1024 threads +0.23 s, 256 or more workgroups of either size +0.24 ... +0.31 s.
    const int want = dst->copy_workgroups > 0 ? dst->copy_workgroups : kWindowCopyBlocks;
    const long long per_grid = want;
    const dim3 cblock(kWindowCopyThreads);
    {
        const long long lo = 0, hi = C;   // (all chains in one dispatch; the kernels take a chain range)
        hipStream_t st = cs;
        if (trace_dst) {
            const long long rows = e->A.cap - e->A.trace_begin;
            const long long row = n * d, sp = rows * d, dp_ = n_out * d;
            const double* src = e->A.trace + lo * sp + (iter_begin - e->A.trace_begin) * d;
            double* out = trace_dst + lo * dp_ + row0 * d;
            const dim3 grid(static_cast<unsigned>(hi - lo < per_grid ? hi - lo : per_grid));
            const bool v2 = row % 2 == 0 && sp % 2 == 0 && dp_ % 2 == 0 && (reinterpret_cast<uintptr_t>(src) % 16 == 0) &&
                            (reinterpret_cast<uintptr_t>(out) % 16 == 0);
            if (v2) { LMC_LAUNCH(window_trace_copy_kernel<2>, grid, cblock, 0, st, src, sp, out, dp_, row, static_cast<int>(hi - lo)); }
            else { LMC_LAUNCH(window_trace_copy_kernel<1>, grid, cblock, 0, st, src, sp, out, dp_, row, static_cast<int>(hi - lo)); }
            HIP_TRY(e, hipGetLastError());
        }
        if (dst->n_planes > 0) {
            const long long total = (hi - lo) * n;
            long long blocks = (total + kWindowCopyThreads - 1) / kWindowCopyThreads;
            if (blocks > per_grid) blocks = per_grid;
            LMC_LAUNCH(window_gather_kernel, dim3(static_cast<unsigned>(blocks)), cblock, 0, st, e->A.stat_rec, e->A.cap,
                       static_cast<int>(lo), static_cast<int>(hi - lo), iter_begin, n_iters, e->cfg.kind == LMC_KIND_HMC ? 1 : 0, W, n_out, row0);
            HIP_TRY(e, hipGetLastError());
        }
    }
    return LMC_OK;
}

int lmc_engine_copy_wait(lmc_engine* e) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    if (!e->copy_stream) return LMC_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->copy_stream));
    return LMC_OK;
}

void* lmc_host_alloc(uint64_t bytes) {
    void* p = nullptr;
    const hipError_t err = hipHostMalloc(&p, bytes > 0 ? static_cast<size_t>(bytes) : 1, hipHostMallocPortable);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        fail(nullptr, LMC_ERR_HIP, "hipHostMalloc(%llu bytes): %s", (unsigned long long)bytes, hipGetErrorString(err));
        return nullptr;
    }
    return p;
}

void lmc_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int lmc_host_register(void* p, uint64_t bytes) {
    if (!p || bytes == 0) return fail(nullptr, LMC_ERR_INVALID, "null argument");
    const hipError_t err = hipHostRegister(p, static_cast<size_t>(bytes), hipHostRegisterPortable | hipHostRegisterMapped);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, LMC_ERR_HIP, "hipHostRegister(%llu bytes): %s", (unsigned long long)bytes, hipGetErrorString(err));
    }
    return LMC_OK;
}

int lmc_host_unregister(void* p) {
    if (!p) return LMC_OK;
    const hipError_t err = hipHostUnregister(p);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, LMC_ERR_HIP, "hipHostUnregister: %s", hipGetErrorString(err));
    }
    return LMC_OK;
}

void* lmc_engine_trace_device_ptr(lmc_engine* e) { return e ? e->A.trace : nullptr; }
void* lmc_engine_stat_records_device_ptr(lmc_engine* e) { return e ? static_cast<void*>(e->A.stat_rec) : nullptr; }
int64_t lmc_engine_trace_begin(lmc_engine* e) { return e ? e->A.trace_begin : 0; }
int64_t lmc_engine_capacity(lmc_engine* e) { return e ? e->A.cap : 0; }

int lmc_engine_get_adapt_state(lmc_engine* e, float* var, double* dual_avg, int32_t* da_count, int32_t* n_samples) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const size_t C = e->cfg.chains, d = e->cfg.dim, dp = e->dpad;
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    if (var)
        HIP_TRY(e, hipMemcpy2D(var, d * sizeof(float), e->A.var, dp * sizeof(float), d * sizeof(float), C, hipMemcpyDefault));
    if (dual_avg) HIP_TRY(e, hipMemcpy(dual_avg, e->A.da, C * 4 * sizeof(double), hipMemcpyDefault));
    if (da_count) HIP_TRY(e, hipMemcpy(da_count, e->A.da_count, C * sizeof(int), hipMemcpyDefault));
    if (n_samples) HIP_TRY(e, hipMemcpy(n_samples, e->A.n_samples, C * sizeof(int), hipMemcpyDefault));
    return LMC_OK;
}

int lmc_engine_keep_moments(lmc_engine* e, int32_t enable) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    ChainArrays& A = e->A;
    const size_t C = e->cfg.chains, dp = e->dpad;
    if (enable && !A.mom_mean) {
        int rc;
        if ((rc = dev_alloc(e, &A.mom_mean, C * dp)) != LMC_OK) return rc;
        if ((rc = dev_alloc(e, &A.mom_m2, C * dp)) != LMC_OK) return rc;
        if ((rc = dev_alloc(e, &A.mom_n, C)) != LMC_OK) return rc;
    } else if (!enable && A.mom_mean) {
        dev_free(e, A.mom_mean); dev_free(e, A.mom_m2); dev_free(e, A.mom_n);
        A.mom_mean = nullptr; A.mom_m2 = nullptr; A.mom_n = nullptr;
    }
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

int lmc_engine_get_moments(lmc_engine* e, double* mean, double* m2, int32_t* n) {
    if (!e) return fail(nullptr, LMC_ERR_INVALID, "null engine");
    if (!e->A.mom_mean) return fail(e, LMC_ERR_STATE, "moments are not kept (call lmc_engine_keep_moments first)");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    const size_t C = e->cfg.chains, d = e->cfg.dim, dp = e->dpad;
    if (mean) HIP_TRY(e, hipMemcpy2D(mean, d * sizeof(double), e->A.mom_mean, dp * sizeof(double), d * sizeof(double), C, hipMemcpyDefault));
    if (m2) HIP_TRY(e, hipMemcpy2D(m2, d * sizeof(double), e->A.mom_m2, dp * sizeof(double), d * sizeof(double), C, hipMemcpyDefault));
    if (n) HIP_TRY(e, hipMemcpy(n, e->A.mom_n, C * sizeof(int), hipMemcpyDefault));
    return LMC_OK;
}

int lmc_engine_get_status(lmc_engine* e, int32_t* status) {
    if (!e || !status) return fail(e, LMC_ERR_INVALID, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    HIP_TRY(e, hipMemcpy(status, e->A.status, e->cfg.chains * sizeof(int), hipMemcpyDefault));
    return LMC_OK;
}

int lmc_engine_get_counters(lmc_engine* e, int64_t* counters) {
    if (!e || !counters) return fail(e, LMC_ERR_INVALID, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    HIP_TRY(e, hipMemcpy(counters, e->A.counters, static_cast<size_t>(e->cfg.chains) * kNumCounters * sizeof(long long),
                         hipMemcpyDefault));
    return LMC_OK;
}

// ---- full chain state (checkpoint / resume / per-iteration parity) ---------------------------------------
static int copy_vec_rows(lmc_engine* e, void* user, void* dev, size_t elem, bool to_user) {
    const size_t C = e->cfg.chains, d = e->cfg.dim, dp = e->dpad;
    if (to_user)
        HIP_TRY(e, hipMemcpy2D(user, d * elem, dev, dp * elem, d * elem, C, hipMemcpyDefault));
    else
        HIP_TRY(e, hipMemcpy2D(dev, dp * elem, user, d * elem, d * elem, C, hipMemcpyDefault));
    return LMC_OK;
}

static int chain_state_xfer(lmc_engine* e, const lmc_chain_state* st, bool to_user) {
    if (!e || !st) return fail(e, LMC_ERR_INVALID, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    ChainArrays& A = e->A;
    const size_t C = e->cfg.chains;
    const size_t plane = C * static_cast<size_t>(e->dpad);
    std::vector<int> wsel(C);
    HIP_TRY(e, hipMemcpy(wsel.data(), A.wsel, C * sizeof(int), hipMemcpyDeviceToHost));
    int rc;
    if (!to_user) {   // canonical layout on load: slot 0 = foreground for every chain
        std::fill(wsel.begin(), wsel.end(), 0);
        if (st->fore_mean || st->fore_raw_var || st->back_mean || st->back_raw_var || st->fore_w_sum || st->back_w_sum) {
            if (!(st->fore_mean && st->fore_raw_var && st->back_mean && st->back_raw_var && st->fore_w_sum && st->back_w_sum))
                return fail(e, LMC_ERR_INVALID, "the Welford fields must be set together");
            HIP_TRY(e, hipMemcpy(A.wsel, wsel.data(), C * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    if (st->var && (rc = copy_vec_rows(e, st->var, A.var, sizeof(float), to_user)) != LMC_OK) return rc;
    if (st->var64) {
        if (!A.var64) return fail(e, LMC_ERR_STATE, "var64 is the state of the general kernels (float64 adaptive diagonal)");
        if ((rc = copy_vec_rows(e, st->var64, A.var64, sizeof(double), to_user)) != LMC_OK) return rc;
    }
    if (!to_user) {
        if (st->fore_mean && (rc = copy_vec_rows(e, st->fore_mean, A.wmean, sizeof(double), false)) != LMC_OK) return rc;
        if (st->fore_raw_var && (rc = copy_vec_rows(e, st->fore_raw_var, A.wraw, sizeof(double), false)) != LMC_OK) return rc;
        if (st->back_mean && (rc = copy_vec_rows(e, st->back_mean, A.wmean + plane, sizeof(double), false)) != LMC_OK) return rc;
        if (st->back_raw_var && (rc = copy_vec_rows(e, st->back_raw_var, A.wraw + plane, sizeof(double), false)) != LMC_OK) return rc;
    } else if (st->fore_mean || st->fore_raw_var || st->back_mean || st->back_raw_var) {
        // chains may sit in different adaptation windows (slot wsel[c] is the foreground): select per chain
        const size_t d = e->cfg.dim, dp = e->dpad;
        std::vector<double> host(2 * plane);
        auto gather = [&](const double* dev, double* fore, double* back) -> int {
            HIP_TRY(e, hipMemcpy(host.data(), dev, 2 * plane * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t c = 0; c < C; ++c) {
                const double* f0 = host.data() + static_cast<size_t>(wsel[c]) * plane + c * dp;
                const double* b0 = host.data() + static_cast<size_t>(1 - wsel[c]) * plane + c * dp;
                if (fore) HIP_TRY(e, hipMemcpy(fore + c * d, f0, d * sizeof(double), hipMemcpyDefault));
                if (back) HIP_TRY(e, hipMemcpy(back + c * d, b0, d * sizeof(double), hipMemcpyDefault));
            }
            return LMC_OK;
        };
        if ((st->fore_mean || st->back_mean) && (rc = gather(A.wmean, st->fore_mean, st->back_mean)) != LMC_OK) return rc;
        if ((st->fore_raw_var || st->back_raw_var) && (rc = gather(A.wraw, st->fore_raw_var, st->back_raw_var)) != LMC_OK) return rc;
    }
    if (to_user && (st->fore_w_sum || st->back_w_sum)) {
        std::vector<double> w(2 * C);
        HIP_TRY(e, hipMemcpy(w.data(), A.wsum, 2 * C * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> fw(C), bw(C);
        for (size_t c = 0; c < C; ++c) { fw[c] = w[2 * c + wsel[c]]; bw[c] = w[2 * c + 1 - wsel[c]]; }
        if (st->fore_w_sum) HIP_TRY(e, hipMemcpy(st->fore_w_sum, fw.data(), C * sizeof(double), hipMemcpyDefault));
        if (st->back_w_sum) HIP_TRY(e, hipMemcpy(st->back_w_sum, bw.data(), C * sizeof(double), hipMemcpyDefault));
    }
    auto strided = [&](double* user, double* dev, int stride, int off) -> int {   // [C] <-> dev[c*stride+off]
        if (!user) return LMC_OK;
        if (to_user)
            HIP_TRY(e, hipMemcpy2D(user, sizeof(double), dev + off, stride * sizeof(double), sizeof(double), C, hipMemcpyDefault));
        else
            HIP_TRY(e, hipMemcpy2D(dev + off, stride * sizeof(double), user, sizeof(double), sizeof(double), C, hipMemcpyDefault));
        return LMC_OK;
    };
    if (!to_user) {
        if ((rc = strided(st->fore_w_sum, A.wsum, 2, 0)) != LMC_OK) return rc;
        if ((rc = strided(st->back_w_sum, A.wsum, 2, 1)) != LMC_OK) return rc;
    }
    if ((rc = strided(st->log_step, A.da, 4, 0)) != LMC_OK) return rc;
    if ((rc = strided(st->log_bar, A.da, 4, 1)) != LMC_OK) return rc;
    if ((rc = strided(st->hbar, A.da, 4, 2)) != LMC_OK) return rc;
    auto ints = [&](int32_t* user, int* dev) -> int {
        if (!user) return LMC_OK;
        if (to_user) HIP_TRY(e, hipMemcpy(user, dev, C * sizeof(int), hipMemcpyDefault));
        else HIP_TRY(e, hipMemcpy(dev, user, C * sizeof(int), hipMemcpyDefault));
        return LMC_OK;
    };
    if ((rc = ints(st->n_samples, A.n_samples)) != LMC_OK) return rc;
    if ((rc = ints(st->da_count, A.da_count)) != LMC_OK) return rc;
    if ((rc = ints(st->iter_count, A.iter_count)) != LMC_OK) return rc;
    if ((rc = ints(st->window, A.awindow)) != LMC_OK) return rc;
    if (!to_user && (st->var || st->var64)) {
        const long long n = static_cast<long long>(C) * e->dpad;
        LMC_LAUNCH(derive_inv_std_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, main_stream(e), e->A, st->var64 ? 1 : 0);
        HIP_TRY(e, hipGetLastError());
        HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    }
    return LMC_OK;
}

int lmc_engine_get_chain_state(lmc_engine* e, const lmc_chain_state* dst) { return chain_state_xfer(e, dst, true); }
int lmc_engine_set_chain_state(lmc_engine* e, const lmc_chain_state* src) { return chain_state_xfer(e, src, false); }

// ---- unit entry points -----------------------------------------------------------------------------------
int lmc_engine_trajectory(lmc_engine* e, const double* q0, const double* p0, int32_t p0_is_f32, double eps,
                          int32_t n_fwd, int32_t n_back, double* out_q, double* out_p, double* out_v, double* out_g,
                          double* out_energy, double* out_logp) {
    if (!e || !q0 || !p0 || !out_q || !out_p || !out_v || !out_g || !out_energy || !out_logp || n_fwd < 0 || n_back < 0)
        return fail(e, LMC_ERR_INVALID, "bad trajectory arguments");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const size_t C = e->cfg.chains, d = e->cfg.dim, ns = static_cast<size_t>(n_fwd + n_back + 1);
    DevBuf<double> dq0, dp0, oq, op, ov, og, oe, ol;
    HIP_TRY(e, dq0.alloc(C * d)); HIP_TRY(e, dp0.alloc(C * d));
    HIP_TRY(e, oq.alloc(C * ns * d)); HIP_TRY(e, op.alloc(C * ns * d));
    HIP_TRY(e, ov.alloc(C * ns * d)); HIP_TRY(e, og.alloc(C * ns * d));
    HIP_TRY(e, oe.alloc(C * ns)); HIP_TRY(e, ol.alloc(C * ns));
    HIP_TRY(e, hipMemcpyAsync(dq0.p, q0, C * d * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipMemcpyAsync(dp0.p, p0, C * d * sizeof(double), hipMemcpyDefault, main_stream(e)));
    if (e->wide) {
        if (e->cfg.target_family == LMC_TARGET_USER && !kUserCompiledInDense) {
            int sdot = e->cfg.start_energy_sdot, p32 = p0_is_f32, nf = n_fwd, nb = n_back;
            double eps_ = eps;
            void* args[] = {&e->A, &e->D, &e->tparams, &dq0.p, &dp0.p, &p32, &sdot, &eps_, &nf, &nb, &oq.p, &op.p, &ov.p, &og.p, &oe.p, &ol.p};
            const int rc = user_launch(e, e->user_trajectory, main_stream(e), static_cast<unsigned>(e->cfg.chains), 64u * e->run_w,
                                       static_cast<unsigned>(e->lds_bytes), args);
            if (rc != LMC_OK) return rc;
        } else {
            const int rc = wide_launch_trajectory(e->cfg.target_family, e->ns, e->run_w, main_stream(e), e->A, e->D, e->tparams, dq0.p, dp0.p,
                                                  p0_is_f32, e->cfg.start_energy_sdot, eps, n_fwd, n_back, oq.p, op.p, ov.p, og.p, oe.p, ol.p);
            if (rc != 0) return dense_fail(e, rc, "trajectory (general kernel)");
        }
    } else if (e->cfg.potential >= LMC_POT_FULL) {
        const int rc = dense_launch_trajectory(e->cfg.target_family, e->ns, pot_f64(e->cfg.potential), main_stream(e),
                                               e->A, e->D, e->tparams, dq0.p, dp0.p, p0_is_f32,
                                               e->cfg.start_energy_sdot, eps, n_fwd, n_back, oq.p, op.p, ov.p, og.p,
                                               oe.p, ol.p);
        if (rc != 0) return dense_fail(e, rc, "trajectory");
    } else {
    const dim3 grid(e->cfg.chains), block(64);
    if (e->cfg.target_family == LMC_TARGET_USER && !kUserCompiledIn) {
        int sdot = e->cfg.start_energy_sdot, p32 = p0_is_f32, nf = n_fwd, nb = n_back;
        double eps_ = eps;
        void* args[] = {&e->A, &e->tparams, &dq0.p, &dp0.p, &p32, &sdot, &eps_, &nf, &nb, &oq.p, &op.p, &ov.p, &og.p, &oe.p, &ol.p};
        const int rc = user_launch(e, e->user_trajectory, main_stream(e), grid.x, block.x, static_cast<unsigned>(e->dpad * 8), args);
        if (rc != LMC_OK) return rc;
    } else {
#define TRAJ_CALL(T)                                                                                          \
    LMC_NS_SWITCH(e, e->ns, LMC_LAUNCH((trajectory_kernel<NS, T>), grid, block, e->dpad * 8, main_stream(e), e->A,   \
                                               e->tparams, dq0.p, dp0.p, p0_is_f32, e->cfg.start_energy_sdot, eps, n_fwd, n_back, oq.p, \
                                               op.p, ov.p, og.p, oe.p, ol.p))
    LMC_FAMILY_SWITCH(e, e->cfg.target_family, TRAJ_CALL)
#undef TRAJ_CALL
    HIP_TRY(e, hipGetLastError());
    }
    }
    HIP_TRY(e, hipMemcpyAsync(out_q, oq.p, C * ns * d * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipMemcpyAsync(out_p, op.p, C * ns * d * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipMemcpyAsync(out_v, ov.p, C * ns * d * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipMemcpyAsync(out_g, og.p, C * ns * d * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipMemcpyAsync(out_energy, oe.p, C * ns * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipMemcpyAsync(out_logp, ol.p, C * ns * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

int lmc_engine_logp_dlogp(lmc_engine* e, const double* q, double* logp, double* grad) {
    if (!e || !q || !logp || !grad) return fail(e, LMC_ERR_INVALID, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const size_t C = e->cfg.chains, d = e->cfg.dim;
    DevBuf<double> dq, dl, dg;
    HIP_TRY(e, dq.alloc(C * d)); HIP_TRY(e, dl.alloc(C)); HIP_TRY(e, dg.alloc(C * d));
    HIP_TRY(e, hipMemcpyAsync(dq.p, q, C * d * sizeof(double), hipMemcpyDefault, main_stream(e)));
    const dim3 grid(e->cfg.chains), block(64);
    if (e->wide && !(e->cfg.target_family == LMC_TARGET_USER && !kUserCompiledIn)) {
        const int rc = wide_launch_logp(e->cfg.target_family, e->ns, e->run_w, main_stream(e), e->A, e->tparams, dq.p, dl.p, dg.p);
        if (rc != 0) return dense_fail(e, rc, "logp_dlogp (general kernel)");
    } else if (e->cfg.target_family == LMC_TARGET_USER && !kUserCompiledIn) {
        void* args[] = {&e->A, &e->tparams, &dq.p, &dl.p, &dg.p};
        const int rc = user_launch(e, e->user_logp, main_stream(e), grid.x, e->wide ? 64u * e->run_w : block.x, e->wide ? 2 * 16 * 8 * 8 : 0, args);
        if (rc != LMC_OK) return rc;
    } else {
#define LOGP_CALL(T) \
    LMC_NS_SWITCH(e, e->ns, LMC_LAUNCH((logp_kernel<NS, T>), grid, block, 0, main_stream(e), e->A, e->tparams, dq.p, dl.p, dg.p))
    LMC_FAMILY_SWITCH(e, e->cfg.target_family, LOGP_CALL)
#undef LOGP_CALL
    HIP_TRY(e, hipGetLastError());
    }
    HIP_TRY(e, hipMemcpyAsync(logp, dl.p, C * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipMemcpyAsync(grad, dg.p, C * d * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

int lmc_engine_rng_draw(lmc_engine* e, const int32_t* ops, int32_t n_ops, double* out) {
    if (!e || !ops || !out || n_ops < 1) return fail(e, LMC_ERR_INVALID, "bad rng_draw arguments");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    std::vector<int32_t> hops(n_ops);
    HIP_TRY(e, hipMemcpy(hops.data(), ops, n_ops * sizeof(int32_t), hipMemcpyDefault));
    long long total = 0, biggest = 2;
    for (int32_t op : hops) {
        total += op > 0 ? op : -op;
        if (op > biggest) biggest = op;
    }
    biggest += 2;
    const size_t C = e->cfg.chains;
    DevBuf<int> dops;
    DevBuf<double> dout, dstage;
    HIP_TRY(e, dops.alloc(n_ops)); HIP_TRY(e, dout.alloc(C * total)); HIP_TRY(e, dstage.alloc(C * biggest));
    HIP_TRY(e, hipMemcpyAsync(dops.p, hops.data(), n_ops * sizeof(int32_t), hipMemcpyHostToDevice, main_stream(e)));
    LMC_LAUNCH(rng_draw_kernel, dim3(e->cfg.chains), dim3(64), 0, main_stream(e), e->A, dops.p, n_ops, dout.p, total, dstage.p, biggest);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipMemcpyAsync(out, dout.p, C * total * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

int lmc_engine_draw_momentum(lmc_engine* e, double* out) {
    if (!e || !out) return fail(e, LMC_ERR_INVALID, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const size_t C = e->cfg.chains, d = e->cfg.dim;
    DevBuf<double> dout;
    HIP_TRY(e, dout.alloc(C * d));
    if (e->wide) {
        const int f32 = e->cfg.potential == LMC_POT_DIAG_ADAPT && !e->cfg.mass_f64;
        const int rc = wide_launch_momentum(e->ns, e->run_w, main_stream(e), e->A, e->D, f32, dout.p);
        if (rc != 0) return dense_fail(e, rc, "draw_momentum (general kernel)");
    } else if (e->cfg.potential >= LMC_POT_FULL) {
        const int rc = dense_launch_momentum(e->ns, main_stream(e), e->A, e->D, dout.p);
        if (rc != 0) return dense_fail(e, rc, "draw_momentum");
    } else {
        const int f32 = e->cfg.potential == LMC_POT_DIAG_ADAPT;
        const dim3 grid(e->cfg.chains), block(64);
        const int lds = 2 * e->dpad * 8;
        LMC_NS_SWITCH(e, e->ns, LMC_LAUNCH((momentum_kernel<NS>), grid, block, lds, main_stream(e), e->A, f32, dout.p))
        HIP_TRY(e, hipGetLastError());
    }
    HIP_TRY(e, hipMemcpyAsync(out, dout.p, C * d * sizeof(double), hipMemcpyDefault, main_stream(e)));
    HIP_TRY(e, hipStreamSynchronize(main_stream(e)));
    return LMC_OK;
}

}  // extern "C"
