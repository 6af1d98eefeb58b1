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
        "peer-to-peer gradient exchange: %u wait(s) for a peer's slice timed out (a rank is missing or stalled); the optimiser steps "
        "from that exchange on were SKIPPED (parameters and moments untouched). ",
        "clip + Adam grid wait: %u workgroup(s) gave up waiting for the rest of the launch (device shared with another process?); "
        "those parameter updates were SKIPPED. ",
        "SAC critic training pass: %u wait(s) for another workgroup's share of q timed out; the affected samples' q, and the critic "
        "gradient of that step, are NaN. "};
    char msg[768] = "";
    uint64_t total = 0;
    for (int s = 0; s < ERL_FAULT_SOURCES; ++s) {
        const uint32_t n = __atomic_load_n(g_fault_host + s, __ATOMIC_ACQUIRE);
        if (!n) continue;
        if (reset) __atomic_store_n(g_fault_host + s, 0u, __ATOMIC_RELEASE);
        if (reset && s == ERL_FAULT_P2P_EXCHANGE) erl_p2p_clear_poison_all();     // the fault is being reported: un-poison the communicators
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
// optional per-launch timing of K6 (measurement hook for bench.py; off by default).  Two clocks on every sampled launch:
//   * a HIP-event bracket on the launch stream.  It contains the dispatch of the kernel behind the first event and the completion
//     signal in front of the second one -- 3 to 15 us on top of the kernel depending on the box (round 3's driver line could not be
//     reconciled with its own step time because of it); erl_k6_timing_null_bracket_us() brackets an EMPTY launch the same way,
//     so that the overhead is a measured number, not a guess;
//   * the kernel's own span on the device's constant-rate clock: thread 0 of every workgroup folds its entry and exit time
//     (wall_clock64) into a {min, max} slot of a library-owned table (Ppo2Args::span) -- first workgroup in to last workgroup out,
//     no host, no command processor in it.  This is what rocprofv3's kernel duration measures to within the dispatch ramp.
// The pairs are kept until erl_k6_timing_read2() drains them.
#include <algorithm>
#include <vector>
static int g_k6_timing = 0;              // 0 = off, n = sample every n-th launch (bracketed) and the launches n / 2 behind them (unbracketed)
static long g_k6_launch = 0;
static bool g_k6_skip = false;
static std::vector<hipEvent_t> g_k6_events;
static hipEvent_t g_k6_open = nullptr;
constexpr int kK6SpanWords = 8, kK6SpanPhases = 7;   // = kSpanWords, kSpanPhases of ppo_step.h: one record per WORKGROUP
constexpr size_t kK6PoolWords = (size_t)4 << 20;     // 32 MiB of records: 2048 sampled launches of 256 workgroups
static unsigned long long *g_k6_pool = nullptr;      // device
static int g_k6_pool_dev = -1;                       // ... the one that was current when the hook was enabled: launches elsewhere are not sampled
static size_t g_k6_pool_used = 0;
struct K6Launch {
    size_t off;          // first word of the launch's records in the pool
    int n_slabs;         // grid = (n_slabs, 2): workgroups [0, n_slabs) are the actor's
    bool bracketed;
    long index;          // the launch's number since erl_k6_timing_enable
};
static std::vector<K6Launch> g_k6_launches;
// of the last erl_k6_timing_read2, [0] over the sampled launches WITHOUT an event bracket around them, [1] over the bracketed ones: shader
// clock inside the launches, a workgroup's own duration, mean shader cycles per phase of an actor workgroup's wave 0 (0: the kernel
// stamps none), summed first-in-to-last-out spans (ms) and their count
struct K6Stats {
    double clock_mhz = 0.0, wg_us = 0.0, phase_cycles[kK6SpanPhases] = {}, span_ms = 0.0;
    int phase_wgs = 0, launches = 0;
};
static K6Stats g_k6_stats[2];
static std::vector<std::pair<long, double>> g_k6_spans[2];      // of the last read: every sampled launch's (number since enable, span in us)
static std::vector<unsigned long long> g_k6_last_records[2];    // raw per-workgroup records of the group's LAST sampled launch (diagnostics)

static void k6_span_reset()
{
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (g_k6_pool && dev != g_k6_pool_dev) {
        (void)hipFree(g_k6_pool);
        g_k6_pool = nullptr;
        g_k6_pool_used = 0;
    }
    g_k6_pool_dev = dev;
    if (!g_k6_pool && hipMalloc((void **)&g_k6_pool, kK6PoolWords * sizeof(unsigned long long)) != hipSuccess) {
        g_k6_pool = nullptr;
        (void)hipGetLastError();
        return;
    }
    if (g_k6_pool_used) (void)hipMemset(g_k6_pool, 0, g_k6_pool_used * sizeof(unsigned long long));
    else (void)hipMemset(g_k6_pool, 0, kK6PoolWords * sizeof(unsigned long long));
    (void)hipDeviceSynchronize();                        // (sampled launches on non-blocking streams are not ordered behind that memset)
    g_k6_pool_used = 0;
    g_k6_launches.clear();
}

// called by erl_ppo_step_f32 right before it enqueues K6 (grid (n_slabs, 2)).  Of every g_k6_timing launches ONE sits inside a HIP-event
// bracket and ONE more (half a period later) is sampled without a bracket; both leave per-workgroup records (ppo_step.h, span_exit:
// plain stores, no atomics), every other launch runs untouched.  Returns the launch's record block (nullptr: not sampled).
unsigned long long *erl_k6_timing_begin(hipStream_t stream, int n_slabs)
{
    if (!g_k6_timing) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != g_k6_pool_dev) return nullptr;    // (the record pool is one device's memory)
    const long k = g_k6_launch++ % g_k6_timing;
    g_k6_skip = k != 0;
    const bool free_sample = g_k6_timing >= 2 && k == g_k6_timing / 2;
    if (g_k6_skip && !free_sample) return nullptr;
    if (!g_k6_skip) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) {
            g_k6_skip = true;
        } else {
            (void)hipEventRecord(e, stream);
            if (g_k6_open) (void)hipEventDestroy(g_k6_open);
            g_k6_open = e;
        }
    }
    const size_t words = (size_t)kK6SpanWords * 2 * (size_t)(n_slabs > 0 ? n_slabs : 0);
    if (!g_k6_pool || !words || g_k6_pool_used + words > kK6PoolWords) return nullptr;
    g_k6_launches.push_back(K6Launch{g_k6_pool_used, n_slabs, !g_k6_skip, g_k6_launch - 1});
    g_k6_pool_used += words;
    return g_k6_pool + g_k6_launches.back().off;
}

// ... and right after
void erl_k6_timing_end(hipStream_t stream)
{
    if (!g_k6_timing || g_k6_skip || !g_k6_open) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, stream);
    g_k6_events.push_back(g_k6_open);
    g_k6_events.push_back(e);
    g_k6_open = nullptr;
}

extern "C" void erl_k6_timing_enable(int every_nth)
{
    g_k6_timing = every_nth > 0 ? every_nth : 0;
    g_k6_launch = 0;
    if (g_k6_timing) k6_span_reset();
}

// waits for the recorded events; returns the summed event-bracket time and the summed in-kernel spans of the SAME (bracketed) launches
// (milliseconds; the latter 0 when the record pool could not be allocated) over `launches` launches, and clears both lists.
extern "C" int erl_k6_timing_read2(double *event_ms, double *span_ms, int *launches)
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
    double span = 0.0;
    g_k6_stats[0] = g_k6_stats[1] = K6Stats{};
    if (g_k6_pool && !g_k6_launches.empty()) {
        int khz = 0;
        const int dev = g_k6_pool_dev;                                   // the clock of the device the records were written on
        (void)hipDeviceSynchronize();
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;   // 100 MHz
        std::vector<unsigned long long> h(g_k6_pool_used);
        if (hipMemcpy(h.data(), g_k6_pool, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int b = 0; b < 2; ++b) {
                double wall = 0.0, mem = 0.0, wgs = 0.0, pwgs = 0.0, ph[kK6SpanPhases] = {}, sp = 0.0;
                int m = 0;
                g_k6_last_records[b].clear();
                g_k6_spans[b].clear();
                for (const K6Launch &L : g_k6_launches) {
                    if ((int)L.bracketed != b) continue;
                    g_k6_last_records[b].assign(h.begin() + L.off, h.begin() + L.off + (size_t)kK6SpanWords * 2 * L.n_slabs);
                    unsigned long long lo = ~0ull, hi = 0ull;
                    for (int w = 0; w < 2 * L.n_slabs; ++w) {
                        const unsigned long long *q = h.data() + L.off + (size_t)kK6SpanWords * w;
                        if (q[1] <= q[0]) continue;          // (a workgroup that left no record)
                        lo = std::min(lo, q[0]);
                        hi = std::max(hi, q[1]);
                        wall += (double)(q[1] - q[0]);
                        mem += (double)(q[3] - q[2]);
                        wgs += 1.0;
                        if (w < L.n_slabs && (q[4] | q[5] | q[6])) {   // an actor workgroup of a kernel that stamps its phases
                            uint32_t prev = (uint32_t)q[2];
                            for (int k = 1; k <= kK6SpanPhases; ++k) {
                                const uint32_t tk = k == kK6SpanPhases ? (uint32_t)q[3] : (uint32_t)(q[4 + ((k - 1) >> 1)] >> (32 * ((k - 1) & 1)));
                                ph[k - 1] += (double)(uint32_t)(tk - prev);
                                prev = tk;
                            }
                            pwgs += 1.0;
                        }
                    }
                    if (hi > lo) {
                        sp += (double)(hi - lo) / khz;
                        ++m;
                        g_k6_spans[b].emplace_back(L.index, (double)(hi - lo) / khz * 1e3);
                    }
                }
                K6Stats &st = g_k6_stats[b];
                st.span_ms = sp;
                st.launches = m;
                // shader cycles per constant-rate tick x the tick rate = the clock the launches ran at
                st.clock_mhz = wall > 0 ? mem / wall * (double)khz * 1e-3 : 0.0;
                st.wg_us = wgs > 0 ? wall / wgs / (double)khz * 1e3 : 0.0;
                st.phase_wgs = (int)pwgs;
                for (int k = 0; k < kK6SpanPhases; ++k) st.phase_cycles[k] = pwgs > 0 ? ph[k] / pwgs : 0.0;
            }
            span = g_k6_stats[1].span_ms;                   // read2's pair: events and spans of the SAME (bracketed) launches
            if (g_k6_stats[1].launches && g_k6_stats[1].launches != n) span *= (double)n / g_k6_stats[1].launches;
        }
        k6_span_reset();
    }
    if (event_ms) *event_ms = tot;
    if (span_ms) *span_ms = span;
    if (launches) *launches = n;
    return ERL_OK;
}

extern "C" int erl_k6_timing_read(double *total_ms, int *launches) { return erl_k6_timing_read2(total_ms, nullptr, launches); }

// what the launches drained by the LAST erl_k6_timing_read2 say (include/erl_hip.h): bracketed = 0: the sampled launches without an
// event bracket (the kernel as the loop runs it), 1: the bracketed ones
extern "C" int erl_k6_timing_clocks(int bracketed, double *span_ms, int *launches, double *shader_mhz, double *workgroup_us,
                                    double *phase_cycles, int max_phases, int *n_phases, int *phase_workgroups)
{
    const K6Stats &st = g_k6_stats[bracketed ? 1 : 0];
    if (span_ms) *span_ms = st.span_ms;
    if (launches) *launches = st.launches;
    if (shader_mhz) *shader_mhz = st.clock_mhz;
    if (workgroup_us) *workgroup_us = st.wg_us;
    const int np = st.phase_wgs > 0 ? kK6SpanPhases : 0;
    if (phase_cycles)
        for (int k = 0; k < np && k < max_phases; ++k) phase_cycles[k] = st.phase_cycles[k];
    if (n_phases) *n_phases = np;
    if (phase_workgroups) *phase_workgroups = st.phase_wgs;
    return ERL_OK;
}

// diagnostics: the raw per-workgroup records (8 u64 each: ppo_step.h, span_exit) of the group's last sampled launch, actor workgroups
// first; returns the number of workgroups copied
extern "C" int erl_k6_timing_last_records(int bracketed, unsigned long long *out, int max_workgroups)
{
    const auto &r = g_k6_last_records[bracketed ? 1 : 0];
    const int n = std::min((int)(r.size() / kK6SpanWords), max_workgroups);
    if (out && n > 0) memcpy(out, r.data(), (size_t)n * kK6SpanWords * sizeof(unsigned long long));
    return n;
}

// every sampled launch of the group drained by the last erl_k6_timing_read2: its number since erl_k6_timing_enable (so a caller that
// knows its loop's length knows where in the loop the launch sat) and its span; returns the number of launches copied
extern "C" int erl_k6_timing_spans(int bracketed, long long *launch_index, double *span_us, int max_launches)
{
    const auto &v = g_k6_spans[bracketed ? 1 : 0];
    const int n = std::min((int)v.size(), max_launches);
    for (int i = 0; i < n; ++i) {
        if (launch_index) launch_index[i] = v[i].first;
        if (span_us) span_us[i] = v[i].second;
    }
    return n;
}

// ---- generic per-kernel spans (erl_common.h: erl_span_slot / erl_span_in / erl_span_out) ----------------------------------------
// one {entry, exit} record per WORKGROUP of a sampled launch, bump-allocated from a device pool; the host folds them (min entry, max
// exit) when a tag is read
constexpr int kSpanTags = ERL_SPAN_TAGS;
constexpr size_t kSpanPoolWords = (size_t)4 << 20;              // 32 MiB: 2 M workgroup records between reads
static unsigned long long *g_span_pool = nullptr;               // device
static size_t g_span_pool_used = 0;
struct SpanLaunch {
    size_t off;
    int64_t n_wg;
    int tag;
};
static std::vector<SpanLaunch> g_span_launches;
static long g_span_count[kSpanTags] = {};
static int g_span_every = 0;                                     // 0 = off, n = every n-th launch of a tag is sampled
static int g_span_dev = -1;                                      // the device the pool lives on: launches on another device leave no record

unsigned long long *erl_span_slot(int tag, int64_t n_workgroups)
{
    if (!g_span_every || !g_span_pool || tag < 0 || tag >= kSpanTags || n_workgroups < 1) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != g_span_dev) return nullptr;      // (the pool is one device's memory)
    if (g_span_count[tag]++ % g_span_every) return nullptr;
    const size_t words = 2 * (size_t)n_workgroups;
    if (g_span_pool_used + words > kSpanPoolWords) return nullptr;
    g_span_launches.push_back(SpanLaunch{g_span_pool_used, n_workgroups, tag});
    g_span_pool_used += words;
    return g_span_pool + g_span_launches.back().off;
}

extern "C" void erl_kernel_span_enable(int every_nth)
{
    g_span_every = every_nth > 0 ? every_nth : 0;
    for (long &c : g_span_count) c = 0;
    if (!g_span_every) return;
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (g_span_pool && dev != g_span_dev) {                      // enabled again from another device: the pool moves with it
        (void)hipFree(g_span_pool);
        g_span_pool = nullptr;
        g_span_pool_used = 0;
    }
    if (!g_span_pool && hipMalloc((void **)&g_span_pool, kSpanPoolWords * sizeof(unsigned long long)) != hipSuccess) {
        g_span_pool = nullptr;
        (void)hipGetLastError();
        return;
    }
    g_span_dev = dev;
    (void)hipDeviceSynchronize();
    (void)hipMemset(g_span_pool, 0, (g_span_pool_used ? g_span_pool_used : kSpanPoolWords) * sizeof(unsigned long long));
    (void)hipDeviceSynchronize();
    g_span_pool_used = 0;
    g_span_launches.clear();
}

// the summed spans (microseconds) of tag's sampled launches since the hook was enabled, and their number; waits for the device.  The
// records stay until the next erl_kernel_span_enable (a tag may be read once per enable).
extern "C" int erl_kernel_span_read(int tag, double *total_us, int *launches)
{
    ERL_REQUIRE(tag >= 0 && tag < kSpanTags, "erl_kernel_span_read: unknown tag %d", tag);
    double tot = 0.0;
    int n = 0;
    if (g_span_pool && g_span_pool_used) {
        int rc = erl_hip_status(hipDeviceSynchronize(), "hipDeviceSynchronize");
        if (rc) return rc;
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, g_span_dev) != hipSuccess || khz <= 0) khz = 100000;
        std::vector<unsigned long long> h;
        for (const SpanLaunch &L : g_span_launches) {
            if (L.tag != tag) continue;
            h.resize(2 * (size_t)L.n_wg);
            rc = erl_hip_status(hipMemcpy(h.data(), g_span_pool + L.off, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost), "hipMemcpy(span records)");
            if (rc) return rc;
            unsigned long long lo = ~0ull, hi = 0ull;
            for (int64_t w = 0; w < L.n_wg; ++w)
                if (h[2 * w + 1] > h[2 * w]) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
            if (hi > lo) { tot += (double)(hi - lo) / khz * 1e3; ++n; }
        }
    }
    if (total_us) *total_us = tot;
    if (launches) *launches = n;
    return ERL_OK;
}

void erl_launch_null_kernel(hipStream_t stream);     // ppo_step.hip

// the event bracket around an EMPTY launch (one workgroup, no work) behind a finished kernel, the way K6 is bracketed in the
// loop: median over `reps` in microseconds -- what a bracket adds to whatever it contains on this box, at this moment.
extern "C" int erl_k6_timing_null_bracket_us(void *stream, int reps, double *median_us)
{
    hipStream_t st = (hipStream_t)stream;
    reps = std::max(1, std::min(reps, 1000));
    std::vector<hipEvent_t> ev(2 * (size_t)reps, nullptr);
    for (auto &e : ev) {
        int rc = erl_hip_status(hipEventCreate(&e), "hipEventCreate");
        if (rc) return rc;
    }
    for (int i = 0; i < reps; ++i) {
        erl_launch_null_kernel(st);                    // "the previous kernel of the loop"
        (void)hipEventRecord(ev[2 * i], st);
        erl_launch_null_kernel(st);
        (void)hipEventRecord(ev[2 * i + 1], st);
    }
    std::vector<double> us;
    for (int i = 0; i < reps; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(ev[2 * i + 1]) == hipSuccess && hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess)
            us.push_back(ms * 1e3);
    }
    for (auto e : ev) (void)hipEventDestroy(e);
    if (us.empty()) { erl_set_error("erl_k6_timing_null_bracket_us: no event pair completed"); return ERL_EINVAL; }
    std::sort(us.begin(), us.end());
    if (median_us) *median_us = us[us.size() / 2];
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
                                 objective, slabs, grads, first_step, lr, beta1, beta2, eps, max_norm, /*adv_stats=*/nullptr,
                                 /*adv_partials=*/nullptr, 0, /*comm=*/nullptr, stream);
}
