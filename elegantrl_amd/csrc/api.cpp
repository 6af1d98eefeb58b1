// Host-only part of liberl_hip.so: ABI version, error string, device query.
#include <stdarg.h>
#include <string.h>

#include "erl_common.h"

static thread_local char g_err[512] = "";

void erl_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int erl_abi_version(void) { return ERL_ABI_VERSION; }

extern "C" const char *erl_last_error_string(void) { return g_err; }

extern "C" int erl_device_info(int *num_cu, int *lds_bytes_per_block)
{
    int dev = 0;
    hipDeviceProp_t prop;
    int rc = erl_hip_status(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    rc = erl_hip_status(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    if (rc) return rc;
    if (num_cu) *num_cu = prop.multiProcessorCount;
    if (lds_bytes_per_block) *lds_bytes_per_block = (int)prop.maxSharedMemoryPerMultiProcessor;
    return ERL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Device-side faults that cannot be returned by the (asynchronous) launch call are counted in a pinned, host-mapped block,
// ONE WORD PER SOURCE: the kernel bumps its word with a system-scope atomic, the host reads them without touching the GPU
// once the stream has been synchronised (erl_async_fault_count).  Allocated on first use; NULL when pinned memory is
// unavailable.
// ---------------------------------------------------------------------------------------------------------
static uint32_t *g_fault_host = nullptr, *g_fault_dev = nullptr;
uint32_t *erl_fault_word(int source)
{
    static bool tried = false;
    if (!tried) {
        tried = true;
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            g_fault_host = (uint32_t *)h;
            g_fault_dev = (uint32_t *)d;
            memset(h, 0, 64);
        } else {
            if (h) (void)hipHostFree(h);
            (void)hipGetLastError();
        }
    }
    return g_fault_dev && source >= 0 && source < ERL_FAULT_SOURCES ? g_fault_dev + source : nullptr;
}

extern "C" int erl_async_fault_count(int reset)
{
    if (!g_fault_host) return 0;
    static const char *const what[ERL_FAULT_SOURCES] = {
        "gae_lookback_kernel: %u look-back wait(s) timed out (a predecessor slab never published); the affected advantages are NaN. ",
        "peer-to-peer gradient exchange: %u wait(s) for a peer's slice timed out (a rank is missing or stalled); the summed gradients "
        "of those minibatches are invalid. ",
        "clip + Adam grid wait: %u workgroup(s) gave up waiting for the rest of the launch (device shared with another process?); "
        "those parameter updates were SKIPPED. "};
    char msg[512] = "";
    uint64_t total = 0;
    for (int s = 0; s < ERL_FAULT_SOURCES; ++s) {
        const uint32_t n = __atomic_load_n(g_fault_host + s, __ATOMIC_ACQUIRE);
        if (!n) continue;
        if (reset) __atomic_store_n(g_fault_host + s, 0u, __ATOMIC_RELEASE);
        total += n;
        const size_t used = strlen(msg);
        snprintf(msg + used, sizeof(msg) - used, what[s], n);
    }
    if (total) erl_set_error("%s", msg);
    return (int)(total > 0x7fffffffull ? 0x7fffffffull : total);
}

// ---------------------------------------------------------------------------------------------------------
// Host-side batching: one C call enqueues a whole PPO update (update_times x [K6, slab reduce, clip + Adam]) so
// that the Python interpreter is off the launch path (about 3 us per launch from here instead of about 10 from
// ctypes).  The loop itself lives in comm.cpp (erl_ppo_update_dp_f32): under data parallelism the gradient all-reduce
// sits between the slab reduction and the optimiser step, issued by RCCL on the same stream.
// ---------------------------------------------------------------------------------------------------------
// optional per-launch timing of K6 (measurement hook for bench.py: erl_ppo_step_f32 brackets its launch with HIP events
// on the launch stream; off by default).  The event pairs are kept until erl_k6_timing_read() drains them.
#include <vector>
static int g_k6_timing = 0;              // 0 = off, n = bracket every n-th launch
static long g_k6_launch = 0;
static bool g_k6_skip = false;
static std::vector<hipEvent_t> g_k6_events;
static hipEvent_t g_k6_open = nullptr;

// called by erl_ppo_step_f32 right before (which = 0) and right after (which = 1) it enqueues K6
void erl_k6_timing_mark(hipStream_t stream, int which)
{
    if (!g_k6_timing) return;
    if (which == 0) g_k6_skip = (g_k6_launch++ % g_k6_timing) != 0;
    if (g_k6_skip) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, stream);
    if (which == 0) {
        if (g_k6_open) (void)hipEventDestroy(g_k6_open);
        g_k6_open = e;
    } else if (g_k6_open) {
        g_k6_events.push_back(g_k6_open);
        g_k6_events.push_back(e);
        g_k6_open = nullptr;
    } else {
        (void)hipEventDestroy(e);
    }
}

extern "C" void erl_k6_timing_enable(int every_nth) { g_k6_timing = every_nth > 0 ? every_nth : 0; g_k6_launch = 0; }

// waits for the recorded events, returns the summed K6 time in milliseconds and the number of launches, and clears.
extern "C" int erl_k6_timing_read(double *total_ms, int *launches)
{
    double tot = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < g_k6_events.size(); i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(g_k6_events[i + 1]) == hipSuccess &&
            hipEventElapsedTime(&ms, g_k6_events[i], g_k6_events[i + 1]) == hipSuccess) {
            tot += ms;
            ++n;
        }
        (void)hipEventDestroy(g_k6_events[i]);
        (void)hipEventDestroy(g_k6_events[i + 1]);
    }
    g_k6_events.clear();
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return ERL_OK;
}

extern "C" int erl_ppo_update_f32(float *flat_params, float *exp_avg, float *exp_avg_sq, const float *act_avg, const float *act_std,
                                  const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A, const float *states,
                                  const float *actions, const uint8_t *unmasks, const float *logprobs, const float *advantages,
                                  const float *reward_sums, int64_t H, int64_t N, const int64_t *ids, int64_t B,
                                  int update_times, float ratio_clip, float lambda_entropy, int objective, float *slabs, float *grads,
                                  int32_t first_step, float lr, float beta1, float beta2, float eps, float max_norm, void *stream)
{
    return erl_ppo_update_dp_f32(flat_params, exp_avg, exp_avg_sq, act_avg, act_std, cri_avg, cri_std, S, h1, h2, A, states, actions,
                                 unmasks, logprobs, advantages, reward_sums, H, N, ids, B, update_times, ratio_clip, lambda_entropy,
                                 objective, slabs, grads, first_step, lr, beta1, beta2, eps, max_norm, /*comm=*/nullptr, stream);
}
