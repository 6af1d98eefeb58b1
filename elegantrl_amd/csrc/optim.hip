// K7: global-norm gradient clip + Adam for up to 4 parameter groups in one launch.  gfx950.
// Replaces clip_grad_norm_ + torch.optim.Adam.step in AgentBase.optimizer_backward
// (elegantrl/agents/AgentBase.py:246-248; Adam built with defaults at AgentPPO.py:24-25).
//
// grid = (blocks_per_group, n_groups).  Each block first computes the group's full squared gradient
// norm on its own (a group is ~25k floats = 100 KB, L2 resident, so the redundant reads are cheaper
// than a second launch or a cross-block handshake, and the summation order is fixed => every block
// derives a bit-identical clip coefficient), then updates its slice.
#include "erl_common.h"

namespace {

struct AdamGroups {
    int64_t off[4], len[4];
};

__global__ __launch_bounds__(1024) void clip_adam_kernel(float *__restrict__ params, const float *__restrict__ grads,
                                                         float *__restrict__ m1, float *__restrict__ m2, AdamGroups gr,
                                                         const int32_t *__restrict__ step_base, int32_t step_offset, float lr,
                                                         float beta1, float beta2, float eps, float max_norm, float grad_scale,
                                                         float host_step_size, float host_bc2_sqrt, float *__restrict__ soft, float tau)
{
    __shared__ double scratch[16];
    const int gi = blockIdx.y;
    const int64_t off = gr.off[gi], len = gr.len[gi];
    const float *g = grads + off;
    // this thread's element of the Adam phase: its four loads ride the same round trip as the norm's
    const int64_t per = (len + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < len) ? lo + per : len;
    const int64_t ie = lo + threadIdx.x;
    const bool own = ie < hi && per <= 1024;          // per > 1024 (very long groups): the loop at the end re-reads
    float e_g = 0.f, e_m1 = 0.f, e_m2 = 0.f, e_p = 0.f;
    if (own) { e_g = g[ie]; e_m1 = m1[off + ie]; e_m2 = m2[off + ie]; e_p = params[off + ie]; }
    double ss = 0.0;
    {   // every load of the thread is issued before the first use: ONE L2 round trip for groups up to 32 Ki elements
        constexpr int U = 32;
        for (int64_t i0 = threadIdx.x; i0 < len; i0 += (int64_t)U * 1024) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + (int64_t)u * 1024;
                x[u] = g[i < len ? i : len - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + (int64_t)u * 1024;
                const float xs = x[u] * grad_scale;
                if (i < len) ss += (double)xs * xs;
            }
        }
    }
    ss = block_sum(ss, scratch);
    const float total_norm = (float)sqrt(ss);
    float coef = max_norm / (total_norm + 1e-6f);   // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
    coef = coef > 1.f ? 1.f : coef;
    const float gmul = grad_scale * coef;

    float step_size = host_step_size, bc2_sqrt = host_bc2_sqrt;   // computed on the host when the step is a host value
    if (step_base) {
        const int step = *step_base + step_offset;
        const double bc1 = 1.0 - pow((double)beta1, (double)step);
        const double bc2 = 1.0 - pow((double)beta2, (double)step);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }

    auto adam = [&](int64_t i, float graw, float m1v, float m2v, float pv) {
        erl_adam_update(erl_mul_rn(graw, gmul), m1v, m2v, pv, beta1, beta2, eps, step_size, bc2_sqrt);
        m1[off + i] = m1v;
        m2[off + i] = m2v;
        params[off + i] = pv;
        // soft target update in the same launch (AgentBase.soft_update :270-278: tar = cur * tau + tar * (1 - tau))
        if (soft) soft[off + i] = erl_soft_update(pv, soft[off + i], tau);
    };
    if (per <= 1024) {
        if (own) adam(ie, e_g, e_m1, e_m2, e_p);
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) adam(i, g[i], m1[off + i], m2[off + i], params[off + i]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Slab reduction + clip + Adam in ONE launch (single-process path: nothing sits between the reduction and the optimiser).
// grid = ceil(stride / 256) workgroups of 1024 threads: element e of the flat gradient is summed by 4 threads in exactly
// grad_reduce_kernel's association (bit-identical gradient), written out, and its square goes into the workgroup's
// per-group partial sum (fp64).  Workgroups then ARRIVE on a monotonic device counter; the one that arrives last -- no
// workgroup ever waits, so no co-residency is assumed and nothing can deadlock -- reads every partial in a fixed order,
// derives the clip coefficients and applies Adam to all parameters (50k elements: 25 per thread and trip, two trips).
// Visibility across CUs / XCDs (cdna_hip_programming.md Guideline 16, form R1): gradient and partials are stored
// write-through (agent-scope relaxed atomic stores = sc1), every storing wave drains its stores before the workgroup's one
// arrival, and the last workgroup makes ONE agent-scope acquire before it reads them back with plain loads.
// MEASURED SLOWER than the two launches it replaces (34.9 vs 18.2 us at config 4, tools/tail_bench.py): the last workgroup's
// Adam phase is one CU moving 1.4 MB (~75 GB/s per CU = ~19 us), which outweighs the saved launch; spreading that phase needs
// every workgroup to WAIT for the norm (a grid barrier, i.e. co-residency assumptions).  Kept as an opt-in entry point
// (ERL_FUSED_TAIL=1 in the update loop) and as the record of why the tail stays two launches.
// ---------------------------------------------------------------------------------------------------------
constexpr int RA_T = 1024, RA_E = 256;      // threads, elements per workgroup

__device__ __forceinline__ void st_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool GRID_WAIT>
__global__ __launch_bounds__(RA_T) void reduce_clip_adam_kernel(const float *__restrict__ slabs, int n_slabs, int64_t stride,
                                                                float *flat, float *__restrict__ params, float *__restrict__ m1,
                                                                float *__restrict__ m2, AdamGroups gr, int n_groups, float lr,
                                                                float beta1, float beta2, float eps, float max_norm, float grad_scale,
                                                                float step_size, float bc2_sqrt, double *partials,
                                                                unsigned *counter, unsigned target, uint32_t *fault)
{
    __shared__ float part[4][RA_E];
    __shared__ double scratch[16];
    __shared__ int s_last;
    const int el = threadIdx.x & (RA_E - 1), p = threadIdx.x / RA_E;
    const int64_t i = (int64_t)blockIdx.x * RA_E + el;
    // (grid-wait form) the element's optimiser state rides the reduction's first round trip
    int my_group = -1;
    float e_m1 = 0.f, e_m2 = 0.f, e_p = 0.f;
    if (GRID_WAIT && p == 0) {
        for (int gi = 0; gi < n_groups; ++gi)
            if (i >= gr.off[gi] && i < gr.off[gi] + gr.len[gi]) my_group = gi;
        if (my_group >= 0) { e_m1 = m1[i]; e_m2 = m2[i]; e_p = params[i]; }
    }
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < stride) {                                  // (the loop nest of grad_reduce_kernel, mlp.hip: same association)
        const float *src = slabs + i;
        int k = p;
        for (; k + 124 < n_slabs; k += 128) {
            float x[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) x[u] = __builtin_nontemporal_load(src + (size_t)(k + 4 * u) * stride);   // streamed once (see mlp.hip)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int u = 0; u < 8; ++u) s[u] += x[8 * v + u];
        }
        for (; k + 28 < n_slabs; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += src[(size_t)(k + 4 * u) * stride];
        }
        for (; k < n_slabs; k += 4) s[0] += src[(size_t)k * stride];
    }
    part[p][el] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    float gsum = 0.f;
    if (p == 0 && i < stride) {
        gsum = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
        st_agent(flat + i, gsum);
    }
    for (int gi = 0; gi < n_groups; ++gi) {            // this workgroup's share of each group's squared norm
        const bool in = p == 0 && i >= gr.off[gi] && i < gr.off[gi] + gr.len[gi];
        const double xs = (double)(gsum * grad_scale);
        const double t = block_sum(in ? xs * xs : 0.0, scratch);
        if (threadIdx.x == 0) __hip_atomic_store(partials + (size_t)blockIdx.x * 4 + gi, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores ...
    __syncthreads();
    if (threadIdx.x == 0) {                            // ... before the workgroup's one arrival
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old + 1u == target);
    }
    __syncthreads();
    if (GRID_WAIT) {
        // ---- every workgroup WAITS for the last arrival, then applies Adam to its own 256 elements from registers: the whole
        // grid is resident (the host checked grid <= the device's capacity for this kernel and the stream is in order, so the
        // workgroups not yet dispatched can only be waiting for unrelated kernels, which finish without us); the spin is
        // bounded all the same and reports through the fault word instead of hanging
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            s_last = 0;                                // reused: 1 = the wait timed out
            while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) {
                    if (fault) __hip_atomic_fetch_add(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    s_last = 1;
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (s_last) return;                            // incomplete norm: the update is SKIPPED (and reported), never applied
        const int nblk = gridDim.x;
        float mul = 0.f;
        for (int gi = 0; gi < n_groups; ++gi) {
            double ss = 0.0;
            for (int b = threadIdx.x; b < nblk; b += RA_T) ss += partials[(size_t)b * 4 + gi];
            ss = block_sum(ss, scratch);
            const float total_norm = (float)sqrt(ss);
            float coef = max_norm / (total_norm + 1e-6f);  // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
            coef = coef > 1.f ? 1.f : coef;
            if (gi == my_group) mul = grad_scale * coef;
        }
        if (my_group >= 0) {
            erl_adam_update(erl_mul_rn(gsum, mul), e_m1, e_m2, e_p, beta1, beta2, eps, step_size, bc2_sqrt);
            m1[i] = e_m1;
            m2[i] = e_m2;
            params[i] = e_p;
        }
        return;
    }
    if (!s_last) return;

    // ---- the last workgroup: ONE agent-scope acquire (drops this CU's stale L1 lines), then plain loads -- the compiler
    // serialises agent-scope atomic loads (52 dependent L2 trips per thread: 34 us for the launch, measured), plain ones it batches
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    // norms in a fixed order, then clip + Adam for every group
    const int nblk = gridDim.x;
    float gmul[4];
    for (int gi = 0; gi < n_groups; ++gi) {
        double ss = 0.0;
        for (int b = threadIdx.x; b < nblk; b += RA_T)
            ss += partials[(size_t)b * 4 + gi];
        ss = block_sum(ss, scratch);
        const float total_norm = (float)sqrt(ss);
        float coef = max_norm / (total_norm + 1e-6f);  // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
        coef = coef > 1.f ? 1.f : coef;
        gmul[gi] = grad_scale * coef;
    }
    for (int gi = 0; gi < n_groups; ++gi) {
        const int64_t off = gr.off[gi], len = gr.len[gi];
        constexpr int U = 13;                          // elements per thread and trip: 4 U loads in flight
        for (int64_t i0 = threadIdx.x; i0 < len; i0 += (int64_t)U * RA_T) {
            float xg[U], xm1[U], xm2[U], xp[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t e = i0 + (int64_t)u * RA_T, ec = e < len ? e : len - 1;
                xg[u] = flat[off + ec];
                xm1[u] = m1[off + ec];
                xm2[u] = m2[off + ec];
                xp[u] = params[off + ec];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t e = i0 + (int64_t)u * RA_T;
                if (e < len) {
                    float a = xm1[u], b = xm2[u], pv = xp[u];
                    erl_adam_update(erl_mul_rn(xg[u], gmul[gi]), a, b, pv, beta1, beta2, eps, step_size, bc2_sqrt);
                    m1[off + e] = a;
                    m2[off + e] = b;
                    params[off + e] = pv;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// clip + Adam for LONG groups (more than 64 Ki elements: SAC's critic ensemble, 280k): clip_adam_kernel makes every
// workgroup read the whole group for the norm, which at 274 workgroups x 1.1 MB is what the launch then costs (17-36 us
// measured in the SAC step).  Here a workgroup squares only its own 1024 elements, publishes the fp64 partial, arrives on
// the device counter and WAITS for the others (whole grid resident -- checked by the host, see reduce_clip_adam_kernel),
// sums the group's partials in a fixed order and updates its elements from registers.  grid = (blocks, n_groups).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void clip_adam_grid_kernel(float *__restrict__ params, const float *__restrict__ grads,
                                                              float *__restrict__ m1, float *__restrict__ m2, AdamGroups gr, float beta1,
                                                              float beta2, float eps, float max_norm, float grad_scale, float step_size,
                                                              float bc2_sqrt, double *partials, unsigned *counter, unsigned target,
                                                              uint32_t *fault, float *__restrict__ soft, float tau)
{
    __shared__ double scratch[16];
    __shared__ int s_timeout;
    const int gi = blockIdx.y;
    const int64_t off = gr.off[gi], len = gr.len[gi];
    const int64_t ie = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const bool own = ie < len;
    float e_g = 0.f, e_m1 = 0.f, e_m2 = 0.f, e_p = 0.f;
    if (own) { e_g = grads[off + ie]; e_m1 = m1[off + ie]; e_m2 = m2[off + ie]; e_p = params[off + ie]; }
    const double xs = (double)(e_g * grad_scale);
    const double part = block_sum(own ? xs * xs : 0.0, scratch);
    if (threadIdx.x == 0) {
        __hip_atomic_store(partials + (size_t)gi * gridDim.x + blockIdx.x, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        s_timeout = 0;
        while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) {
                if (fault) __hip_atomic_fetch_add(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_timeout = 1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (s_timeout) return;                             // incomplete norm: this workgroup's update is SKIPPED (and reported)
    double ss = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 1024) ss += partials[(size_t)gi * gridDim.x + b];
    ss = block_sum(ss, scratch);
    const float total_norm = (float)sqrt(ss);
    float coef = max_norm / (total_norm + 1e-6f);   // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
    coef = coef > 1.f ? 1.f : coef;
    if (own) {
        erl_adam_update(erl_mul_rn(e_g, erl_mul_rn(grad_scale, coef)), e_m1, e_m2, e_p, beta1, beta2, eps, step_size, bc2_sqrt);
        m1[off + ie] = e_m1;
        m2[off + ie] = e_m2;
        params[off + ie] = e_p;
        if (soft) soft[off + ie] = erl_soft_update(e_p, soft[off + ie], tau);
    }
}

// clip + Adam from squared-norm PIECES the gradient's producer left (sac_fused.hip dw_table_kernel: one fp64 sum per weight-gradient
// workgroup): every workgroup adds the `nparts` pieces in one fixed order -- no pass over the gradient, no arrival counter, no wait --
// and updates its 1024 elements; the soft target update rides along as in clip_adam_grid_kernel.  One group (the whole optimiser).
__global__ __launch_bounds__(1024) void clip_adam_parts_kernel(float *__restrict__ params, const float *__restrict__ grads,
                                                               float *__restrict__ m1, float *__restrict__ m2, int64_t len,
                                                               const double *__restrict__ parts, int nparts, float beta1, float beta2,
                                                               float eps, float max_norm, float step_size, float bc2_sqrt,
                                                               float *__restrict__ soft, float tau)
{
    __shared__ double scratch[16];
    const int64_t ie = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const bool own = ie < len;
    const int64_t ic = own ? ie : len - 1;
    float e_g = grads[ic], e_m1 = m1[ic], e_m2 = m2[ic], e_p = params[ic];       // (one round trip with the pieces)
    const float e_s = soft ? soft[ic] : 0.f;
    double ss = 0.0;
    for (int b = threadIdx.x; b < nparts; b += 1024) ss += parts[b];
    ss = block_sum(ss, scratch);
    const float total_norm = (float)sqrt(ss);
    float coef = max_norm / (total_norm + 1e-6f);   // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
    coef = coef > 1.f ? 1.f : coef;
    if (own) {
        erl_adam_update(erl_mul_rn(e_g, erl_mul_rn(1.0f, coef)), e_m1, e_m2, e_p, beta1, beta2, eps, step_size, bc2_sqrt);
        m1[ie] = e_m1;
        m2[ie] = e_m2;
        params[ie] = e_p;
        if (soft) soft[ie] = erl_soft_update(e_p, e_s, tau);
    }
}

struct RaScratch {
    char *ptr = nullptr;        // [counter (256 B)][partials: kRaMaxBlocks x 4 doubles]
    unsigned base = 0;
};
constexpr int kRaMaxBlocks = 8192;
static RaScratch g_ra[32];

}  // namespace

// the per-device arrival counter and partial-norm table of the single-launch tails, and the counter value the launch that
// is about to be enqueued completes at (`nblk` arrivals after everything enqueued before it).  One table per device: these
// launches are meant for ONE stream per device (the update loop's); launches from concurrent streams would share the counter.
static int ra_arrivals(const char *what, int64_t nblk, hipStream_t st, RaScratch **out, unsigned *target)
{
    int dev = -1;
    ERL_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 32, "%s: no device", what);
    RaScratch &sc = g_ra[dev];
    if (!sc.ptr) {
        void *ptr = nullptr;
        const size_t bytes = 256 + (size_t)kRaMaxBlocks * 4 * sizeof(double);
        int rc = erl_hip_status(hipMalloc(&ptr, bytes), "hipMalloc(arrival scratch)");
        if (rc) return rc;
        if ((rc = erl_hip_status(hipMemset(ptr, 0, bytes), "hipMemset(arrival scratch)"))) return rc;
        if ((rc = erl_hip_status(hipDeviceSynchronize(), "hipDeviceSynchronize(arrival scratch)"))) return rc;    // (once: a non-blocking stream is not ordered behind that memset)
        sc.ptr = (char *)ptr;
        sc.base = 0;
    }
    if ((uint64_t)sc.base + (uint64_t)nblk >= 0x7fffff00ull) {      // the arrival counter is about to wrap: restart it behind earlier work
        int rc = erl_hip_status(hipMemsetAsync(sc.ptr, 0, 256, st), "hipMemsetAsync(arrival counter)");
        if (rc) return rc;
        sc.base = 0;
    }
    *target = sc.base + (unsigned)nblk;
    sc.base = *target;
    *out = &sc;
    return ERL_OK;
}

static int device_capacity(const void *kernel, int threads)
{
    int dev = -1, per_cu = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return per_cu * cus > 0 ? per_cu * cus : -1;
}

static int reduce_clip_adam_launch(bool grid_wait, const float *slabs, int n_slabs, int64_t stride, float *flat_grad, float *params,
                                   float *exp_avg, float *exp_avg_sq, const int64_t *group_off, const int64_t *group_len,
                                   int n_groups, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                                   float grad_scale, void *stream)
{
    ERL_REQUIRE(slabs && flat_grad && params && exp_avg && exp_avg_sq && group_off && group_len, "erl_reduce_clip_adam_f32: NULL argument");
    ERL_REQUIRE(n_slabs >= 1 && stride >= 1 && n_groups >= 1 && n_groups <= 4 && step >= 1, "erl_reduce_clip_adam_f32: bad argument");
    AdamGroups gr;
    for (int i = 0; i < 4; ++i) {
        gr.off[i] = i < n_groups ? group_off[i] : 0;
        gr.len[i] = i < n_groups ? group_len[i] : 0;
        ERL_REQUIRE(gr.off[i] >= 0 && gr.len[i] >= 0 && gr.off[i] + gr.len[i] <= stride, "erl_reduce_clip_adam_f32: group outside the gradient row");
    }
    const int64_t nblk = erl_cdiv(stride, RA_E);
    ERL_REQUIRE(nblk <= kRaMaxBlocks, "erl_reduce_clip_adam_f32: gradient row too long (%lld floats)", (long long)stride);
    RaScratch *scp = nullptr;
    unsigned target = 0;
    hipStream_t st = (hipStream_t)stream;
    {
        int rc = ra_arrivals("erl_reduce_clip_adam_f32", nblk, st, &scp, &target);
        if (rc) return rc;
    }
    RaScratch &sc = *scp;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    if (grid_wait)
        hipLaunchKernelGGL(reduce_clip_adam_kernel<true>, dim3((unsigned)nblk), dim3(RA_T), 0, st, slabs, n_slabs, stride, flat_grad, params,
                           exp_avg, exp_avg_sq, gr, n_groups, lr, beta1, beta2, eps, max_norm, grad_scale, (float)((double)lr / bc1),
                           (float)sqrt(bc2), reinterpret_cast<double *>(sc.ptr + 256), reinterpret_cast<unsigned *>(sc.ptr), target,
                           erl_fault_word(ERL_FAULT_ADAM_GRID_WAIT));
    else
        hipLaunchKernelGGL(reduce_clip_adam_kernel<false>, dim3((unsigned)nblk), dim3(RA_T), 0, st, slabs, n_slabs, stride, flat_grad, params,
                           exp_avg, exp_avg_sq, gr, n_groups, lr, beta1, beta2, eps, max_norm, grad_scale, (float)((double)lr / bc1),
                           (float)sqrt(bc2), reinterpret_cast<double *>(sc.ptr + 256), reinterpret_cast<unsigned *>(sc.ptr), target,
                           (uint32_t *)nullptr);
    ERL_LAUNCH_CHECK("erl_reduce_clip_adam_f32");
}

// 1 when the grid-wait form of erl_reduce_clip_adam_f32 may be used for a gradient row of `stride` floats on the current
// device: every workgroup of the launch fits on the device at once (occupancy of the kernel x compute units)
extern "C" int erl_reduce_clip_adam_grid_ok(int64_t stride)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 0;
    static int capacity[32] = {0};
    if (!capacity[dev]) capacity[dev] = device_capacity((const void *)reduce_clip_adam_kernel<true>, RA_T);
    return capacity[dev] > 0 && erl_cdiv(stride, RA_E) <= capacity[dev];
}

static int clip_adam_impl(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, const int64_t *group_off,
                          const int64_t *group_len, int n_groups, const int32_t *step_base, int32_t step_offset, float lr, float beta1,
                          float beta2, float eps, float max_norm, float grad_scale, float *soft, float tau, void *stream)
{
    ERL_REQUIRE(params && grads && exp_avg && exp_avg_sq && group_off && group_len, "erl_clip_adam_f32: NULL argument");
    ERL_REQUIRE(n_groups >= 1 && n_groups <= 4, "erl_clip_adam_f32: n_groups must be 1..4");
    ERL_REQUIRE(step_base || step_offset >= 1, "erl_clip_adam_f32: Adam step must be >= 1");
    AdamGroups gr;
    int64_t longest = 0;
    for (int i = 0; i < 4; ++i) {
        gr.off[i] = i < n_groups ? group_off[i] : 0;
        gr.len[i] = i < n_groups ? group_len[i] : 0;
        ERL_REQUIRE(gr.off[i] >= 0 && gr.len[i] >= 0, "erl_clip_adam_f32: negative group bounds");
        if (gr.len[i] > longest) longest = gr.len[i];
    }
    int bx = (int)erl_cdiv(longest, 1024);   // one element per thread in the Adam phase (latency bound: no serial loop)
    if (bx < 1) bx = 1;
    if (bx > 64 && !step_base) {             // long groups: partial norms + a grid-wide wait, if the whole launch is resident at once
        int dev = -1;
        static int capacity[32] = {0};
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 32) {
            if (!capacity[dev]) capacity[dev] = device_capacity((const void *)clip_adam_grid_kernel, 1024);
            if ((int64_t)bx * n_groups <= capacity[dev] && (int64_t)bx * n_groups <= kRaMaxBlocks * 4) {
                RaScratch *sc = nullptr;
                unsigned target = 0;
                int rc = ra_arrivals("erl_clip_adam_f32", (int64_t)bx * n_groups, (hipStream_t)stream, &sc, &target);
                if (rc) return rc;
                const double bc1 = 1.0 - pow((double)beta1, (double)step_offset), bc2 = 1.0 - pow((double)beta2, (double)step_offset);
                hipLaunchKernelGGL(clip_adam_grid_kernel, dim3(bx, n_groups), dim3(1024), 0, (hipStream_t)stream, params, grads, exp_avg,
                                   exp_avg_sq, gr, beta1, beta2, eps, max_norm, grad_scale, (float)((double)lr / bc1), (float)sqrt(bc2),
                                   reinterpret_cast<double *>(sc->ptr + 256), reinterpret_cast<unsigned *>(sc->ptr), target, erl_fault_word(ERL_FAULT_ADAM_GRID_WAIT),
                                   soft, tau);
                ERL_LAUNCH_CHECK("erl_clip_adam_f32");
            }
        }
    }
    if (bx > 64) bx = 64;
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (!step_base) {
        const double bc1 = 1.0 - pow((double)beta1, (double)step_offset), bc2 = 1.0 - pow((double)beta2, (double)step_offset);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
    hipLaunchKernelGGL(clip_adam_kernel, dim3(bx, n_groups), dim3(1024), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       gr, step_base, step_offset, lr, beta1, beta2, eps, max_norm, grad_scale, step_size, bc2_sqrt, soft, tau);
    ERL_LAUNCH_CHECK("erl_clip_adam_f32");
}

extern "C" int erl_clip_adam_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, const int64_t *group_off,
                                 const int64_t *group_len, int n_groups, const int32_t *step_base, int32_t step_offset, float lr,
                                 float beta1, float beta2, float eps, float max_norm, float grad_scale, void *stream)
{
    return clip_adam_impl(params, grads, exp_avg, exp_avg_sq, group_off, group_len, n_groups, step_base, step_offset, lr, beta1, beta2, eps,
                          max_norm, grad_scale, nullptr, 0.f, stream);
}

// the same with the soft target update  soft <- params_new * tau + soft * (1 - tau)  of every updated element folded into the
// launch (AgentBase.soft_update, elegantrl/agents/AgentBase.py:270-278; the arithmetic of soft_update_kernel, sac.hip);
// `soft` has the layout of `params` and may be NULL.  Internal (sac_fused.hip).
int erl_clip_adam_soft_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, const int64_t *group_off,
                           const int64_t *group_len, int n_groups, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                           float grad_scale, float *soft, float tau, hipStream_t stream)
{
    return clip_adam_impl(params, grads, exp_avg, exp_avg_sq, group_off, group_len, n_groups, nullptr, step, lr, beta1, beta2, eps, max_norm,
                          grad_scale, soft, tau, (void *)stream);
}


// clip_grad_norm_ + Adam (+ soft target update) of ONE group whose squared gradient norm arrives as `nparts` fp64 pieces (see
// clip_adam_parts_kernel).  Internal (sac_fused.hip).
int erl_clip_adam_parts_soft_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t len, const double *parts,
                                 int nparts, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm, float *soft, float tau,
                                 hipStream_t stream)
{
    ERL_REQUIRE(params && grads && exp_avg && exp_avg_sq && parts, "erl_clip_adam_parts_soft_f32: NULL tensor");
    ERL_REQUIRE(len >= 1 && nparts >= 1 && step >= 1, "erl_clip_adam_parts_soft_f32: bad shape len=%lld nparts=%d step=%d", (long long)len, nparts, (int)step);
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(clip_adam_parts_kernel, dim3((unsigned)erl_cdiv(len, 1024)), dim3(1024), 0, stream, params, grads, exp_avg, exp_avg_sq, len,
                       parts, nparts, beta1, beta2, eps, max_norm, (float)((double)lr / bc1), (float)sqrt(bc2), soft, tau);
    ERL_LAUNCH_CHECK("erl_clip_adam_parts_soft_f32");
}

extern "C" int erl_reduce_clip_adam_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, float *params,
                                        float *exp_avg, float *exp_avg_sq, const int64_t *group_off, const int64_t *group_len,
                                        int n_groups, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                                        float grad_scale, void *stream)
{
    return reduce_clip_adam_launch(false, slabs, n_slabs, stride, flat_grad, params, exp_avg, exp_avg_sq, group_off, group_len, n_groups, step,
                                   lr, beta1, beta2, eps, max_norm, grad_scale, stream);
}

// The same in the grid-wait form (every workgroup waits for the norm and updates its own elements); EINVAL when the launch
// would not fit on the device at once (erl_reduce_clip_adam_grid_ok).
extern "C" int erl_reduce_clip_adam_grid_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, float *params,
                                             float *exp_avg, float *exp_avg_sq, const int64_t *group_off, const int64_t *group_len,
                                             int n_groups, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                                             float grad_scale, void *stream)
{
    ERL_REQUIRE(erl_reduce_clip_adam_grid_ok(stride), "erl_reduce_clip_adam_grid_f32: %lld-float row does not fit the device in one wave of workgroups",
                (long long)stride);
    return reduce_clip_adam_launch(true, slabs, n_slabs, stride, flat_grad, params, exp_avg, exp_avg_sq, group_off, group_len, n_groups, step,
                                   lr, beta1, beta2, eps, max_norm, grad_scale, stream);
}
