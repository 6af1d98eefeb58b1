// The optimiser tail of a PPO minibatch (elegantrl/agents/AgentBase.py:239-248 after backward()) in TWO launches, with the
// data-parallel exchange inside the first one:
//
//   1  reduce_exchange_kernel   sums the per-workgroup gradient slabs of K6 (grad_reduce_kernel's association, bit for bit);
//                               under data parallelism every workgroup then PUSHES its 256 reduced elements into its row of
//                               every peer's stage (remote stores over xGMI, own rank: a local store), publishes a per-workgroup
//                               sequence flag to every peer, waits for the `world` flags of its own slice (it polls LOCAL
//                               memory only) and forms the rank-ordered sum of the rows from its own memory; writes the flat
//                               gradient and the workgroup's fp64 share of every parameter group's squared norm
//   2  clip_adam_partials_kernel sums <= ceil(stride / 256) partial norms in a fixed order (not 25k gradients per workgroup as
//                               clip_adam_kernel does), clips, applies Adam to its own elements.
//
// No workgroup waits on another workgroup of its OWN launch (it publishes before it waits, and what it waits for is the same
// slice of the peers' launches), so no co-residency is assumed; the wait is bounded and reports through
// erl_async_fault_count.  Stage reuse: rows are double buffered by the parity of the sequence number; a rank overwrites half
// (s & 1) in launch s + 2, which it enters only after launch s + 1 finished, whose workgroups waited for every peer's s + 1
// flags, which a peer publishes only after ITS launch s has finished reading -- no second handshake.
// Every rank adds the same rows in the same (rank) order: the replicas' gradients, hence weights, stay bit-identical.
// The same kernel with one "slab" is the standalone all-reduce (erl_comm_allreduce_sum_f32 / _f64 on a p2p communicator)
// and the squared-norm pass after a foreign all-reduce (RCCL / torch.distributed routes: erl_grad_sq_partials_f32).
#include "erl_common.h"
#include "s3_image.h"
#include "ppo_step_wd.h"

namespace {

constexpr int kMaxPartialChunks = 32768;     // 64-element chunks the partial-norm table holds (rows up to 2 Mi floats)

struct TailGroups {
    int64_t off[4], len[4];
};
// a ONE-GROUP launch of the slab reduction (two-chain update loop, comm.cpp): group >= 0: the grid covers blocks blk0 .. blk0 + n_own - 1 (the
// group's elements) and then the blocks from blk_logs on (the logged sums behind the last group); -1: every block, every group
struct TailPart {
    int group = -1;
    int n_own = 0;
    int64_t blk0 = 0, blk_logs = 0;
};

// the slabs are read once (ERL_SLAB_LD, A/B builds only: 1 plain, 2 `sc1`)
#ifndef ERL_SLAB_LD
#define ERL_SLAB_LD 0
#endif
template <typename T>
__device__ __forceinline__ T ld_stream(const T *p)
{
#if ERL_SLAB_LD == 0
    return __builtin_nontemporal_load(p);
#elif ERL_SLAB_LD == 1
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// System-coherent accesses spelled out (sc0 sc1 = write through to / read from memory, past L1 and L2, local or over xGMI)
// instead of C++ system-scope fences: a release / acquire fence at system scope is a whole-L2 write-back / invalidate PER
// WAVE (buffer_wbl2 / buffer_inv sc1), 3184 of them per launch here -- measured 73 us for the 203 KB exchange against 9 us
// for the slab reduction alone.  The stage is uncached memory and these accesses bypass the caches by themselves, so the
// ordering that is needed is only: my stores are ACKNOWLEDGED (s_waitcnt vmcnt(0) in every storing wave) before the
// workgroup barrier that precedes the flag store; the rows are read after the barrier that follows the flag poll.
__device__ __forceinline__ void st_sys(float *p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sys(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void ld_sys_issue(float &d, const float *p) { asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void ld_sys_issue(double &d, const double *p) { asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(d) : "v"(p) : "memory"); }
template <typename T>
__device__ __forceinline__ void ld_sys_wait(T (&x)[ERL_P2P_MAX_WORLD])     // the loads above have landed; ties their registers to the wait
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"memory");
}

// NT threads reduce NT / 4 elements (4 threads per element, grad_reduce_kernel's association whatever NT is).  The partial
// norms are per 64-element CHUNK (= the wave of threads that holds the chunk's sums: one butterfly, no cross-wave step), so
// every block shape writes the same table bit for bit.
template <typename T, bool DP, int NT>
__global__ __launch_bounds__(NT) void reduce_exchange_kernel(const T *slabs, int n_slabs, int64_t stride, T *out, TailGroups gr,
                                                             int n_groups, float grad_scale, double *partials, ErlExchange ex,
                                                             unsigned long long *span = nullptr, TailPart tp = TailPart{})
{
    const unsigned long long t_span = erl_span_in(span);
    constexpr int NE = NT / 4;
    __shared__ T part[4][NE];
    const int el = threadIdx.x & (NE - 1), p = threadIdx.x / NE;
    // (a one-group launch of the two-chain update loop covers the group's blocks, then the blocks of the logged sums behind the last group)
    const int64_t blk = tp.group < 0 ? (int64_t)blockIdx.x : ((int)blockIdx.x < tp.n_own ? tp.blk0 + (int64_t)blockIdx.x : tp.blk_logs + ((int64_t)blockIdx.x - tp.n_own));
    const int64_t i = blk * NE + el;
    T s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < stride) {                                  // (the loop nest of grad_reduce_kernel, mlp.hip: same association)
        const T *src = slabs + i;
        int k = p;
        for (; k + 124 < n_slabs; k += 128) {          // 32 loads in flight: one round trip per 128 slabs
            T x[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) x[u] = ld_stream(src + (size_t)(k + 4 * u) * stride);   // slabs are streamed once
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
    T gsum = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);        // all four threads of an element hold it
    if (DP) {
        // ---- a peer whose wait timed out in an earlier launch of this update loop has raised ITS word in my table: its replica no longer
        // steps, so neither may mine (round 4 poisoned the timed-out rank only: the others kept stepping and the replicas diverged
        // silently until that rank raised).  One workgroup looks; the sticky device word stops clip + Adam, the fault word makes the host raise.
        if (blockIdx.x == 0 && (int)threadIdx.x < ex.world && ex.poison) {
            const uint32_t *pw = ex.flags[ex.rank] + (int64_t)ex.world * ex.nblk_max + threadIdx.x;
            if (__hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) &&
                !__hip_atomic_fetch_or(ex.poison, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) && ex.fault)
                __hip_atomic_fetch_add(ex.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // ---- push my reduced slice into my row of every rank's stage (thread quarter p serves ranks p, p + 4)
        const int64_t half_off = (int64_t)(ex.seq & 1u) * ex.half_bytes;
        if (i < stride)
            for (int r = p; r < ex.world; r += 4)
                st_sys(reinterpret_cast<T *>(ex.stage[r] + half_off + (int64_t)ex.rank * ex.row_bytes) + i, gsum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its stores are acknowledged by their memories ...
        __syncthreads();
        if ((int)threadIdx.x < ex.world) {             // ... before the slice's flag is raised on every rank (myself included)
            __hip_atomic_store(ex.flags[threadIdx.x] + (int64_t)ex.rank * ex.nblk_max + blockIdx.x, ex.seq, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t *f = ex.flags[ex.rank] + (int64_t)threadIdx.x * ex.nblk_max + blockIdx.x;     // my own memory
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - ex.seq) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > ex.spin_limit) {
                    // a peer never arrived: reported, never a hang.  The sum below is then garbage, so the communicator is POISONED
                    // (sticky device word): clip_adam_partials_kernel leaves parameters, moments and W2 images untouched for the
                    // rest of the update loop -- the same policy as a timed-out grid wait in optim.hip -- and the error is raised
                    // when the host reads the fault counter at the end of update_net (which also clears the word)
                    if (ex.fault) __hip_atomic_fetch_add(ex.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (ex.poison) {
                        __hip_atomic_fetch_or(ex.poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // tell every rank (their next launch reads it): no replica steps once one of them has stopped
                        for (int r = 0; r < ex.world; ++r)
                            __hip_atomic_store(ex.flags[r] + (int64_t)ex.world * ex.nblk_max + ex.rank, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
            }
        }
        __syncthreads();
        if (p == 0 && i < stride) {                    // the rank-ordered sum of the rows, all from my own memory
            const char *rows = ex.stage[ex.rank] + half_off;
            T x[ERL_P2P_MAX_WORLD];
#pragma unroll
            for (int r = 0; r < ERL_P2P_MAX_WORLD; ++r) {
                x[r] = (T)0;
                if (r < ex.world) ld_sys_issue(x[r], reinterpret_cast<const T *>(rows + (int64_t)r * ex.row_bytes) + i);
            }
            ld_sys_wait(x);
            gsum = x[0];
#pragma unroll
            for (int r = 1; r < ERL_P2P_MAX_WORLD; ++r)
                if (r < ex.world) gsum += x[r];        // rank order on every rank: bit-identical sums
        }
    }
    if (tp.group >= 0) {
        // one-group launch: this chain owns its group's elements and, behind the last group, the logged sums its network's workgroups
        // write (ppo_step kernels: [critic objective, actor objective, entropy, 0, zero pad ...] -- entries 1 and 2 are the actor's);
        // everything else in the blocks it covers belongs to the other chain, whose slabs may be half written right now
        const int64_t lb = gr.off[n_groups - 1] + gr.len[n_groups - 1];
        const bool mine = i >= lb ? ((i - lb == 1 || i - lb == 2) ? tp.group == 0 : tp.group == n_groups - 1)
                                  : (i >= gr.off[tp.group] && i < gr.off[tp.group] + gr.len[tp.group]);
        if (p == 0 && i < stride && mine) out[i] = gsum;
        if (partials && threadIdx.x < NE) {
            const double xs = (double)((float)gsum * grad_scale), sq = xs * xs;
            const bool in = i < stride && i >= gr.off[tp.group] && i < gr.off[tp.group] + gr.len[tp.group];
            const double t = wave_sum(in ? sq : 0.0);          // (the bits of the all-groups launch: its general path, or a sum over a chunk that lies inside the group)
            if ((threadIdx.x & 63) == 0) partials[(size_t)(i >> 6) * 4 + tp.group] = t;
        }
        erl_span_out(span, t_span);
        return;
    }
    if (p == 0 && i < stride) out[i] = gsum;
    if (partials && threadIdx.x < NE) {                // whole waves (NE is a multiple of 64): chunk c = elements 64 c .. 64 c + 63
        const int64_t chunk = i >> 6, c_lo = chunk << 6, c_hi = c_lo + 63;
        const double xs = (double)((float)gsum * grad_scale), sq = xs * xs;
        // a chunk almost always lies inside ONE group (the only straddler at config 4 is the chunk holding the actor | critic
        // seam): one butterfly then, the other groups' partials are exactly 0 -- the same bits the general path produces
        int own = -1, touched = 0;
        for (int gi = 0; gi < n_groups; ++gi) {
            const bool overlap = c_lo < gr.off[gi] + gr.len[gi] && c_hi >= gr.off[gi];
            const bool inside = c_lo >= gr.off[gi] && c_hi < gr.off[gi] + gr.len[gi];
            touched += overlap;
            if (inside) own = gi;
        }
        if (touched == 0 || (touched == 1 && own >= 0)) {          // (wave-uniform: depends on the chunk only)
            const double t = own >= 0 ? wave_sum(i < stride ? sq : 0.0) : 0.0;
            if ((threadIdx.x & 63) == 0)
                for (int gi = 0; gi < n_groups; ++gi) partials[(size_t)chunk * 4 + gi] = gi == own ? t : 0.0;
        } else {
            for (int gi = 0; gi < n_groups; ++gi) {
                const bool in = i < stride && i >= gr.off[gi] && i < gr.off[gi] + gr.len[gi];
                const double t = wave_sum(in ? sq : 0.0);
                if ((threadIdx.x & 63) == 0) partials[(size_t)chunk * 4 + gi] = t;
            }
        }
    }
    erl_span_out(span, t_span);
}

// clip + Adam on ONE element (index ie of parameter group gi at `off`) + the refresh of the split-arithmetic minibatch kernels' weight
// images that follow their fp32 weights (s3_image.h): the arithmetic of every tail of this file
__device__ __forceinline__ void tail_adam_apply(float e_g, float e_m1, float e_m2, float e_p, float coef, float grad_scale, float beta1, float beta2,
                                                float eps, float step_size, float bc2_sqrt, float *__restrict__ params, float *__restrict__ m1,
                                                float *__restrict__ m2, int64_t off, int64_t ie, int gi, const S3Images &im)
{
    erl_adam_update(erl_mul_rn(e_g, erl_mul_rn(grad_scale, coef)), e_m1, e_m2, e_p, beta1, beta2, eps, step_size, bc2_sqrt);
    m1[off + ie] = e_m1;
    m2[off + ie] = e_m2;
    params[off + ie] = e_p;
    if (gi < 2 && im.net[gi].img) {
        const int64_t e = ie - im.net[gi].w2_off;
        const int h1 = im.net[gi].h1, S = im.net[gi].S;
        if (e >= 0 && e < (int64_t)h1 * im.net[gi].h2) s3_image_put_w2(im.net[gi].img, h1, im.net[gi].h2, (int)(e / h1), (int)(e % h1), e_p);
        else if (im.net[gi].img1 && ie < (int64_t)h1 * S) s3_image_put(im.net[gi].img1, im.net[gi].K1, (int)(ie / S), (int)(ie % S), e_p);
    }
}

// clip + Adam from the partial norms: grid = (ceil(longest / 1024), n_groups); one element per thread, its four loads
// and the partials ride one round trip.
__global__ __launch_bounds__(1024) void clip_adam_partials_kernel(float *__restrict__ params, const float *__restrict__ grads,
                                                                  float *__restrict__ m1, float *__restrict__ m2, TailGroups gr,
                                                                  const double *__restrict__ partials, int nblk, float beta1,
                                                                  float beta2, float eps, float max_norm, float grad_scale,
                                                                  float step_size, float bc2_sqrt, S3Images im,
                                                                  const uint32_t *__restrict__ poison, unsigned long long *span, int g0 = 0,
                                                                  bool own_chunks_only = false)
{
    __shared__ double scratch[16];
    const unsigned long long t_span = erl_span_in(span);
    // a gradient exchange of this update loop timed out (reduce_exchange_kernel): its sums are garbage -- touch nothing
    if (poison && __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const int gi = blockIdx.y + g0;
    const int64_t off = gr.off[gi], len = gr.len[gi];
    const int64_t ie = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const bool own = ie < len;
    float e_g = 0.f, e_m1 = 0.f, e_m2 = 0.f, e_p = 0.f;
    if (own) { e_g = grads[off + ie]; e_m1 = m1[off + ie]; e_m2 = m2[off + ie]; e_p = params[off + ie]; }
    double ss = 0.0;
    if (own_chunks_only) {
        // behind a one-group slab reduction: only the chunks that overlap the group were written; the others are exact zeros after an
        // all-groups reduction, so skipping them leaves the same sum in the same association
        const int64_t c0 = off >> 6, c1 = (off + len - 1) >> 6;
        for (int b = threadIdx.x; b < nblk; b += 1024) ss += (b >= c0 && b <= c1) ? partials[(size_t)b * 4 + gi] : 0.0;
    } else {
        for (int b = threadIdx.x; b < nblk; b += 1024) ss += partials[(size_t)b * 4 + gi];
    }
    ss = block_sum(ss, scratch);
    const float total_norm = (float)sqrt(ss);
    float coef = max_norm / (total_norm + 1e-6f);      // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
    coef = coef > 1.f ? 1.f : coef;
    if (own) tail_adam_apply(e_g, e_m1, e_m2, e_p, coef, grad_scale, beta1, beta2, eps, step_size, bc2_sqrt, params, m1, m2, off, ie, gi, im);
    erl_span_out(span, t_span);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The tail as ONE launch (round 5; single process): reduce_exchange_kernel's slab sum and partial norms, then -- instead of a kernel
// boundary and clip_adam_partials_kernel -- every workgroup waits until ALL workgroups' partial norms are published, sums them in
// clip_adam_partials_kernel's order (the same bits) and applies clip + Adam to the 256 elements it has just reduced, from registers.
// No arrival counter and no flags: the partial norms ARE the flags.  The table is double buffered by the launch's parity; an entry not
// yet written holds a sentinel (a NaN pattern no sum of squares produces); workgroup b of launch k (parity p) resets ITS entries of
// parity 1 - p -- last read by launch k - 1, which has finished -- so launch k + 1 finds them blank.  Entries are published by
// agent-scope stores and polled by agent-scope loads (they bypass the XCD-private L2s), 1 .. 2 chunks per thread, all in flight at
// once; what the last poll returned is what is summed: one round trip after the slowest workgroup has published.  Round 2's
// single-launch tails counted arrivals on ONE device counter (199 same-address atomics, an acquire fence per workgroup, then another
// pass over the table): 16.0 us back to back against this kernel's ~9.  The grid (ceil(stride / 256) workgroups of 1024 threads: 199 at
// config 4) must be resident at once: the host checks it against the device's capacity; the wait is bounded all the same, and a
// timeout SKIPS the update and reports through erl_async_fault_count (the policy of every bounded wait of this library).  The skip is
// decided PER WORKGROUP: a workgroup that saw every partial norm applies clip + Adam to its 256 elements, one whose wait expired does not,
// so after a fault the parameters and moments are PARTIALLY stepped -- the host must treat the fault as fatal for this optimiser state
// (restore a checkpoint), not carry on; AgentPPO raises at its next host sync.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr unsigned long long kTailBlank = 0x7ff8dead0000beefull;      // quiet NaN with a payload: never the result of a sum of squares
struct FusedTail {
    float *params, *m1, *m2;
    float beta1, beta2, eps, max_norm, step_size, bc2_sqrt;
    S3Images im;
    double *cur, *other;          // partial-norm tables [chunk][4]: this launch's parity / the other one (blanked here)
    int nchunks;                  // 64-element chunks of the row
    uint32_t spin_limit;
    uint32_t *fault;
};

__global__ __launch_bounds__(1024) void tail_fused_kernel(const float *slabs, int n_slabs, int64_t stride, float *out, TailGroups gr, int n_groups,
                                                          float grad_scale, FusedTail ft, unsigned long long *span)
{
    constexpr int NT = 1024, NE = NT / 4, MAXR = 2;
    __shared__ float part[4][NE];
    __shared__ double scratch[16];
    const unsigned long long t_span = erl_span_in(span);
    const int el = threadIdx.x & (NE - 1), p = threadIdx.x / NE;
    const int64_t i = (int64_t)blockIdx.x * NE + el;
    // the element's optimiser state rides the reduction's first round trip (quarter p == 0 owns the elements)
    int my_group = -1;
    float e_m1 = 0.f, e_m2 = 0.f, e_p = 0.f;
    if (p == 0) {
        for (int gi = 0; gi < n_groups; ++gi)
            if (i >= gr.off[gi] && i < gr.off[gi] + gr.len[gi]) my_group = gi;
        if (my_group >= 0) { e_m1 = ft.m1[i]; e_m2 = ft.m2[i]; e_p = ft.params[i]; }
    }
    if (threadIdx.x < 16) {                            // blank this workgroup's four chunks of the OTHER parity
        const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 2);
        if (c < ft.nchunks)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(ft.other) + c * 4 + (threadIdx.x & 3), kTailBlank, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < stride) {                                  // (the loop nest of grad_reduce_kernel, mlp.hip: same association)
        const float *src = slabs + i;
        int k = p;
        for (; k + 124 < n_slabs; k += 128) {          // 32 loads in flight: one round trip per 128 slabs
            float x[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) x[u] = ld_stream(src + (size_t)(k + 4 * u) * stride);   // slabs are streamed once
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
    const float gsum = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
    if (p == 0 && i < stride) out[i] = gsum;
    if (threadIdx.x < NE) {                            // whole waves: chunk c = elements 64 c .. 64 c + 63 (reduce_exchange_kernel's partials)
        const int64_t chunk = i >> 6, c_lo = chunk << 6, c_hi = c_lo + 63;
        const double xs = (double)(gsum * grad_scale), sq = xs * xs;
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(ft.cur) + (size_t)chunk * 4;
        int own = -1, touched = 0;
        for (int gi = 0; gi < n_groups; ++gi) {
            const bool overlap = c_lo < gr.off[gi] + gr.len[gi] && c_hi >= gr.off[gi];
            const bool inside = c_lo >= gr.off[gi] && c_hi < gr.off[gi] + gr.len[gi];
            touched += overlap;
            if (inside) own = gi;
        }
        if (touched == 0 || (touched == 1 && own >= 0)) {          // (wave-uniform: depends on the chunk only)
            const double t = own >= 0 ? wave_sum(i < stride ? sq : 0.0) : 0.0;
            if ((threadIdx.x & 63) == 0 && chunk < ft.nchunks)
                for (int gi = 0; gi < n_groups; ++gi)
                    __hip_atomic_store(dst + gi, (unsigned long long)__double_as_longlong(gi == own ? t : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (int gi = 0; gi < n_groups; ++gi) {
                const bool in = i < stride && i >= gr.off[gi] && i < gr.off[gi] + gr.len[gi];
                const double t = wave_sum(in ? sq : 0.0);
                if ((threadIdx.x & 63) == 0 && chunk < ft.nchunks)
                    __hip_atomic_store(dst + gi, (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- wait until every chunk's partial norms are there: thread t polls chunks t, t + 1024 (rows up to 131 072 floats)
    const unsigned long long *cur = reinterpret_cast<const unsigned long long *>(ft.cur);
    unsigned long long v[MAXR][4];
    uint32_t spins = 0;
    bool timed_out = false;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int b = (int)threadIdx.x + NT * r;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                v[r][gi] = 0ull;                       // (+0.0: what a chunk beyond the row contributes)
                if (b < ft.nchunks && gi < n_groups) {
                    v[r][gi] = __hip_atomic_load(cur + (size_t)b * 4 + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && v[r][gi] != kTailBlank;
                }
            }
        }
        if (__syncthreads_and(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > ft.spin_limit) {                 // (block-uniform: every thread counts the same rounds)
            timed_out = true;
            break;
        }
    }
    if (timed_out) {                                   // incomplete norm: THIS workgroup's elements are not updated (and the fault is reported)
        if (threadIdx.x == 0 && ft.fault) __hip_atomic_fetch_add(ft.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        erl_span_out(span, t_span);
        return;
    }
    // ---- clip_adam_partials_kernel's sums, group by group (thread-strided partials, then the block sum: the same bits)
    float coef_mine = 1.f;
    for (int gi = 0; gi < n_groups; ++gi) {
        double ss = 0.0;
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((int)threadIdx.x + NT * r < ft.nchunks) ss += __longlong_as_double((long long)v[r][gi]);
        ss = block_sum(ss, scratch);
        const float total_norm = (float)sqrt(ss);
        float coef = ft.max_norm / (total_norm + 1e-6f);      // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
        coef = coef > 1.f ? 1.f : coef;
        if (gi == my_group) coef_mine = coef;
    }
    if (my_group >= 0)
        tail_adam_apply(gsum, e_m1, e_m2, e_p, coef_mine, grad_scale, ft.beta1, ft.beta2, ft.eps, ft.step_size, ft.bc2_sqrt, ft.params, ft.m1, ft.m2,
                        gr.off[my_group], i - gr.off[my_group], my_group, ft.im);
    erl_span_out(span, t_span);
}

// W2 and W1 images of both networks from the flat parameters [actor | critic]: one thread per image element (W1's pad columns: 0)
__global__ __launch_bounds__(256) void s3_image_build_kernel(const float *__restrict__ params, int64_t Pa, S3Images im,
                                                             const double *__restrict__ adv_partials, int n_partials, int H, int N,
                                                             double *__restrict__ adv_stats, S3AuxSrc ax, float *__restrict__ aux)
{
    if (blockIdx.y == 3) {          // the per-sample records (s3_image.h): one thread per buffer row, exact copies
        const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (aux && r < ax.rows) {
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = j < ax.A ? ax.actions[r * ax.A + j] : 0.f;
            float4 *dst = reinterpret_cast<float4 *>(aux + r * kS3AuxFloats);
            dst[0] = make_float4(a[0], a[1], a[2], a[3]);
            dst[1] = make_float4(a[4], a[5], a[6], a[7]);
            dst[2] = make_float4(ax.logprobs[r], ax.advantages[r], ax.reward_sums[r], ax.unmasks[r] ? 1.f : 0.f);
            dst[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    if (blockIdx.y == 2) {          // the advantage statistics' fold rides this launch (one block)
        __shared__ double scratch[4];
        if (blockIdx.x == 0 && adv_partials) erl_adv_stats_fold_block(adv_partials, n_partials, H, N, adv_stats, scratch);
        return;
    }
    const int gi = blockIdx.y;
    const int h1 = im.net[gi].h1, S = im.net[gi].S, K1 = im.net[gi].K1;
    const float *P = params + (gi ? Pa : 0);
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n2 = (int64_t)h1 * im.net[gi].h2;
    if (e < n2) {
        s3_image_put_w2(im.net[gi].img, h1, im.net[gi].h2, (int)(e / h1), (int)(e % h1), P[im.net[gi].w2_off + e]);
    } else if (im.net[gi].img1 && e - n2 < (int64_t)h1 * K1) {
        const int row = (int)((e - n2) / K1), col = (int)((e - n2) % K1);
        s3_image_put(im.net[gi].img1, K1, row, col, col < S ? P[(int64_t)row * S + col] : 0.f);
    }
}

// the three logged objectives of update_net (AgentPPO.py:168-171: means over the minibatches) from the gradient rows' tails:
// out[j] = scale * mean_k rows[k][offset + j], j < 3, summed in row order by one thread each -- replaces a torch mean + mul pair
__global__ __launch_bounds__(192) void logs_mean_kernel(const float *__restrict__ rows, int64_t stride, int64_t offset, int n_rows, float scale,
                                                        float *__restrict__ out)
{
    // 64 row slots x 3 columns: the loads of up to 64 rows are in flight together (one thread per column walking 40 rows 203 KB
    // apart took 11 us); the slots meet in LDS in a fixed order
    __shared__ float part[64][3];
    const int slot = threadIdx.x / 3, j = threadIdx.x - 3 * slot;
    float s = 0.f;
    for (int k = slot; k < n_rows; k += 64) s += rows[(size_t)k * stride + offset + j];
    part[slot][j] = s;
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
        for (int q = 0; q < 64; ++q) t += part[q][threadIdx.x];
        out[threadIdx.x] = t / (float)n_rows * scale;
    }
}

// the partial-norm table of an optimiser tail: library-owned, one per (device, stream) that ran one -- written by launch 1 and
// read by launch 2 of the SAME stream, so two update loops on different streams of a device never share a table
struct PartialsSlot {
    int device = -1;
    hipStream_t stream = nullptr;
    double *buf = nullptr;
};
PartialsSlot g_partials[32];

int fill_groups(const char *what, const int64_t *off, const int64_t *len, int n_groups, int64_t stride, TailGroups *gr)
{
    ERL_REQUIRE(n_groups >= 0 && n_groups <= 4 && (n_groups == 0 || (off && len)), "%s: n_groups must be 0..4", what);
    for (int i = 0; i < 4; ++i) {
        gr->off[i] = i < n_groups ? off[i] : 0;
        gr->len[i] = i < n_groups ? len[i] : 0;
        ERL_REQUIRE(gr->off[i] >= 0 && gr->len[i] >= 0 && gr->off[i] + gr->len[i] <= stride, "%s: group outside the gradient row", what);
    }
    return ERL_OK;
}


// ---- the single-launch tail (tail_fused_kernel): per (device, stream, row length) a double-buffered partial-norm table, blank at first
struct FusedSlot {
    int device = -1;
    hipStream_t stream = nullptr;
    int64_t stride = 0;
    double *buf = nullptr;        // [2 parities][nchunks_alloc][4]
    int64_t nchunks_alloc = 0;
    unsigned parity = 0;
    int capacity = -1;            // workgroups of tail_fused_kernel the device holds at once
};
FusedSlot g_fused[32];

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long *p, unsigned long long v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}

}  // namespace

// 1 when the single-launch tail can serve a gradient row of `stride` floats on the current device: the row fits the kernel's two
// chunks per thread and every workgroup of the launch is resident at once (the kernel waits for ALL of its workgroups)
extern "C" int erl_tail_fused_ok(int64_t stride)
{
    if (stride < 1 || erl_cdiv(stride, 64) > 2048) return 0;
    int dev = -1, per_cu = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)tail_fused_kernel, 1024, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return (int64_t)per_cu * cus >= erl_cdiv(stride, 256) ? 1 : 0;
}

// slabs (n_slabs, stride) -> out (stride) AND clip + Adam on `params` (groups off / len of the row) AND the weight images' refresh: ONE launch
int erl_tail_fused_f32(const float *slabs, int n_slabs, int64_t stride, float *out, const int64_t *off, const int64_t *len, int n_groups,
                       float grad_scale, float *params, float *exp_avg, float *exp_avg_sq, int32_t step, float lr, float beta1, float beta2,
                       float eps, float max_norm, const S3Images *images, hipStream_t stream)
{
    ERL_REQUIRE(slabs && out && params && exp_avg && exp_avg_sq && n_slabs >= 1 && stride >= 1 && step >= 1, "single-launch tail: bad argument");
    ERL_REQUIRE(n_groups >= 1, "single-launch tail: no parameter group");
    TailGroups gr;
    int rc = fill_groups("single-launch tail", off, len, n_groups, stride, &gr);
    if (rc) return rc;
    int dev = -1;
    ERL_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0, "single-launch tail: no device");
    const int64_t nchunks = erl_cdiv(stride, 64);
    ERL_REQUIRE(nchunks <= 2048, "single-launch tail: row too long (%lld floats)", (long long)stride);
    FusedSlot *slot = nullptr;
    for (auto &f : g_fused)
        if (f.buf && f.device == dev && f.stream == stream && f.stride == stride) { slot = &f; break; }
    if (!slot) {
        for (auto &f : g_fused)
            if (!f.buf) { slot = &f; break; }
        ERL_REQUIRE(slot, "single-launch tail: more than 32 (device, stream, row length) triples");
        const int64_t alloc = erl_cdiv(stride, 256) * 4;              // whole workgroups: 4 chunks each
        void *pbuf = nullptr;
        if ((rc = erl_hip_status(hipMalloc(&pbuf, (size_t)2 * alloc * 4 * sizeof(double)), "hipMalloc(single-launch tail table)"))) return rc;
        hipLaunchKernelGGL(fill_u64_kernel, dim3(64), dim3(256), 0, stream, (unsigned long long *)pbuf, kTailBlank, 2 * alloc * 4);
        slot->buf = (double *)pbuf;
        slot->device = dev; slot->stream = stream; slot->stride = stride; slot->nchunks_alloc = alloc; slot->parity = 0;
    }
    FusedTail ft{};
    ft.params = params; ft.m1 = exp_avg; ft.m2 = exp_avg_sq;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    ft.beta1 = beta1; ft.beta2 = beta2; ft.eps = eps; ft.max_norm = max_norm;
    ft.step_size = (float)((double)lr / bc1); ft.bc2_sqrt = (float)sqrt(bc2);
    ft.im = images ? *images : S3Images{};
    ft.cur = slot->buf + (size_t)(slot->parity & 1u) * slot->nchunks_alloc * 4;
    ft.other = slot->buf + (size_t)((slot->parity & 1u) ^ 1u) * slot->nchunks_alloc * 4;
    slot->parity++;
    ft.nchunks = (int)nchunks;
    static const uint32_t spin = [] { const char *e = getenv("ERL_TAIL_SPIN"); return e && atol(e) > 0 ? (uint32_t)atol(e) : (1u << 22); }();
    ft.spin_limit = spin;
    ft.fault = erl_fault_word(ERL_FAULT_ADAM_GRID_WAIT);
    const int64_t nblk = erl_cdiv(stride, 256);
    hipLaunchKernelGGL(tail_fused_kernel, dim3((unsigned)nblk), dim3(1024), 0, stream, slabs, n_slabs, stride, out, gr, n_groups, grad_scale, ft,
                       erl_span_slot(ERL_SPAN_SLAB_REDUCE, nblk));
    ERL_LAUNCH_CHECK("single-launch tail");
}

extern "C" int erl_reduce_clip_adam_fused_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, float *params, float *exp_avg,
                                              float *exp_avg_sq, const int64_t *group_off, const int64_t *group_len, int n_groups, int32_t step,
                                              float lr, float beta1, float beta2, float eps, float max_norm, float grad_scale, void *stream)
{
    ERL_REQUIRE(erl_tail_fused_ok(stride), "erl_reduce_clip_adam_fused_f32: a %lld-float row cannot take the single-launch tail on this device "
                "(erl_tail_fused_ok)", (long long)stride);
    return erl_tail_fused_f32(slabs, n_slabs, stride, flat_grad, group_off, group_len, n_groups, grad_scale, params, exp_avg, exp_avg_sq, step, lr,
                              beta1, beta2, eps, max_norm, nullptr, (hipStream_t)stream);
}

int erl_tail_partials(double **out, hipStream_t stream)
{
    int dev = -1;
    ERL_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0, "gradient tail: no device");
    PartialsSlot *slot = nullptr;
    for (auto &s : g_partials)
        if (s.buf && s.device == dev && s.stream == stream) { slot = &s; break; }
    if (!slot) {
        for (auto &s : g_partials)
            if (!s.buf) { slot = &s; break; }
        ERL_REQUIRE(slot, "gradient tail: more than 32 (device, stream) pairs ran an optimiser tail");
        void *p = nullptr;
        int rc = erl_hip_status(hipMalloc(&p, (size_t)kMaxPartialChunks * 4 * sizeof(double)), "hipMalloc(partial norms)");
        if (rc) return rc;
        slot->buf = (double *)p;
        slot->device = dev;
        slot->stream = stream;
    }
    *out = slot->buf;
    return ERL_OK;
}

// slabs (n_slabs, stride) -> out (stride) [+ exchange over `ex`] [+ partial norms of `n_groups` groups]
int erl_launch_reduce_exchange_f32(const float *slabs, int n_slabs, int64_t stride, float *out, const int64_t *off, const int64_t *len,
                                   int n_groups, float grad_scale, bool want_partials, const ErlExchange *ex, hipStream_t stream)
{
    ERL_REQUIRE(slabs && out && n_slabs >= 1 && stride >= 1, "gradient reduce / exchange: bad argument");
    TailGroups gr;
    int rc = fill_groups("gradient reduce / exchange", off, len, n_groups, stride, &gr);
    if (rc) return rc;
    double *partials = nullptr;
    if (want_partials) {
        // (the exchanging form writes whole 256-element workgroups: up to 4 * ceil(stride / 256) chunks)
        ERL_REQUIRE(4 * erl_cdiv(stride, 256) <= kMaxPartialChunks, "gradient reduce / exchange: row too long for the partial-norm table (%lld floats)",
                    (long long)stride);
        if ((rc = erl_tail_partials(&partials, stream))) return rc;
    }
    if (ex) {       // 1024 threads x 256 elements: one flag per workgroup and peer -- 199 x world small remote stores per exchange at config 4
        const int64_t nblk = erl_cdiv(stride, 256);
        ERL_REQUIRE(nblk <= ex->nblk_max && stride * (int64_t)sizeof(float) <= ex->row_bytes,
                    "gradient exchange: %lld floats > the %lld the peer stages were sized for", (long long)stride, (long long)(ex->row_bytes / 4));
        hipLaunchKernelGGL((reduce_exchange_kernel<float, true, 1024>), dim3((unsigned)nblk), dim3(1024), 0, stream, slabs, n_slabs, stride, out, gr,
                           n_groups, grad_scale, partials, *ex, erl_span_slot(ERL_SPAN_SLAB_REDUCE, nblk));
    } else {        // 256 threads x 64 elements (grad_reduce_kernel's shape: 795 workgroups keep every CU's memory pipes busy)
        hipLaunchKernelGGL((reduce_exchange_kernel<float, false, 256>), dim3((unsigned)erl_cdiv(stride, 64)), dim3(256), 0, stream, slabs, n_slabs,
                           stride, out, gr, n_groups, grad_scale, partials, ErlExchange{}, erl_span_slot(ERL_SPAN_SLAB_REDUCE, erl_cdiv(stride, 64)));
    }
    ERL_LAUNCH_CHECK("gradient reduce / exchange");
}

// the slab reduction of ONE parameter group (two-chain update loop, comm.cpp): its elements of the gradient row, its partial norms, and the
// logged sums behind the last group that its network's workgroups wrote -- the bits the all-groups launch leaves in those places
int erl_launch_reduce_group_f32(const float *slabs, int n_slabs, int64_t stride, float *out, const int64_t *off, const int64_t *len, int n_groups,
                                int group, float grad_scale, hipStream_t stream)
{
    ERL_REQUIRE(slabs && out && n_slabs >= 1 && stride >= 1 && n_groups >= 1 && group >= 0 && group < n_groups, "gradient reduce (one group): bad argument");
    TailGroups gr;
    int rc = fill_groups("gradient reduce (one group)", off, len, n_groups, stride, &gr);
    if (rc) return rc;
    ERL_REQUIRE(gr.len[group] >= 1 && 4 * erl_cdiv(stride, 256) <= kMaxPartialChunks, "gradient reduce (one group): bad group / row too long");
    double *partials = nullptr;
    if ((rc = erl_tail_partials(&partials, stream))) return rc;
    TailPart tp;
    tp.group = group;
    tp.blk0 = gr.off[group] / 64;
    const int64_t blk1 = erl_cdiv(gr.off[group] + gr.len[group], 64);              // one past the group's last block
    tp.n_own = (int)(blk1 - tp.blk0);
    const int64_t lb0 = (gr.off[n_groups - 1] + gr.len[n_groups - 1]) / 64, lb1 = erl_cdiv(stride, 64);     // blocks of the logged sums
    tp.blk_logs = lb0 >= tp.blk0 && lb0 < blk1 ? blk1 : lb0;                         // (those not covered already)
    const int64_t n_logs = lb1 > tp.blk_logs ? lb1 - tp.blk_logs : 0;
    const int64_t nblk = tp.n_own + n_logs;
    hipLaunchKernelGGL((reduce_exchange_kernel<float, false, 256>), dim3((unsigned)nblk), dim3(256), 0, stream, slabs, n_slabs, stride, out, gr, n_groups,
                       grad_scale, partials, ErlExchange{}, erl_span_slot(ERL_SPAN_SLAB_REDUCE, nblk), tp);
    ERL_LAUNCH_CHECK("gradient reduce (one group)");
}

int erl_launch_exchange_f64(double *buf, int64_t count, const ErlExchange *ex, hipStream_t stream)
{
    ERL_REQUIRE(buf && ex && count >= 1, "erl_comm_allreduce_sum_f64: bad argument");
    const int64_t nblk = erl_cdiv(count, 256);
    ERL_REQUIRE(nblk <= ex->nblk_max && count * (int64_t)sizeof(double) <= ex->row_bytes,
                "erl_comm_allreduce_sum_f64: %lld doubles > the %lld the peer stages were sized for", (long long)count, (long long)(ex->row_bytes / 8));
    hipLaunchKernelGGL((reduce_exchange_kernel<double, true, 1024>), dim3((unsigned)nblk), dim3(1024), 0, stream, (const double *)buf, 1, count, buf,
                       TailGroups{}, 0, 1.f, (double *)nullptr, *ex);
    ERL_LAUNCH_CHECK("erl_comm_allreduce_sum_f64");
}

extern "C" int erl_ppo_logs_mean_f32(const float *grad_rows, int64_t stride, int64_t offset, int n_rows, float scale, float *out3, void *stream)
{
    ERL_REQUIRE(grad_rows && out3 && n_rows >= 1 && offset >= 0 && offset + 3 <= stride, "erl_ppo_logs_mean_f32: bad argument");
    hipLaunchKernelGGL(logs_mean_kernel, dim3(1), dim3(192), 0, (hipStream_t)stream, grad_rows, stride, offset, n_rows, scale, out3);
    ERL_LAUNCH_CHECK("erl_ppo_logs_mean_f32");
}

// the last launch of update_net when the advantages came from the rollout's epilogue: the logged means as above (workgroup 0) AND
// the side effect of get_advantages on the caller's buffers (elegantrl/agents/AgentPPO.py:211-214): rewards[trunc] += values[trunc],
// undones[trunc] = False -- one fp32 add per truncated element, as gae_exact_kernel's `mutate` does
namespace {
__global__ __launch_bounds__(256) void ppo_finish_kernel(const float *__restrict__ rows, int64_t stride, int64_t offset, int n_rows, float scale,
                                                         float *__restrict__ out, float *__restrict__ rewards, uint8_t *__restrict__ undones,
                                                         const uint8_t *__restrict__ unmasks, const float *__restrict__ values, int64_t total)
{
    if (blockIdx.x == 0) {
        __shared__ float part[64][3];
        if (threadIdx.x < 192) {
            const int slot = threadIdx.x / 3, j = threadIdx.x - 3 * slot;
            float s = 0.f;
            for (int k = slot; k < n_rows; k += 64) s += rows[(size_t)k * stride + offset + j];
            part[slot][j] = s;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            float t = 0.f;
            for (int q = 0; q < 64; ++q) t += part[q][threadIdx.x];
            out[threadIdx.x] = t / (float)n_rows * scale;
        }
        return;
    }
    const int64_t stride_e = (int64_t)(gridDim.x - 1) * 256;
    for (int64_t i = (int64_t)(blockIdx.x - 1) * 256 + threadIdx.x; i < total; i += stride_e) {
        if (!unmasks[i]) {
            rewards[i] = erl_add_rn(rewards[i], values[i]);
            undones[i] = 0;
        }
    }
}
}  // namespace

extern "C" int erl_ppo_finish_f32(const float *grad_rows, int64_t stride, int64_t offset, int n_rows, float scale, float *out3, float *rewards,
                                  uint8_t *undones, const uint8_t *unmasks, const float *values, int64_t total, void *stream)
{
    ERL_REQUIRE(grad_rows && out3 && n_rows >= 1 && offset >= 0 && offset + 3 <= stride, "erl_ppo_finish_f32: bad argument");
    ERL_REQUIRE(rewards && undones && unmasks && values && total >= 1, "erl_ppo_finish_f32: NULL rollout tensor");
    int nblk = (int)erl_cdiv(total, 256 * 4);
    if (nblk > 512) nblk = 512;
    hipLaunchKernelGGL(ppo_finish_kernel, dim3(1 + (unsigned)nblk), dim3(256), 0, (hipStream_t)stream, grad_rows, stride, offset, n_rows, scale, out3,
                       rewards, undones, unmasks, values, total);
    ERL_LAUNCH_CHECK("erl_ppo_finish_f32");
}

extern "C" int erl_grad_reduce_partials_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, const int64_t *group_off,
                                            const int64_t *group_len, int n_groups, float grad_scale, void *stream)
{
    return erl_launch_reduce_exchange_f32(slabs, n_slabs, stride, flat_grad, group_off, group_len, n_groups, grad_scale, true, nullptr,
                                          (hipStream_t)stream);
}

extern "C" int erl_grad_sq_partials_f32(float *grads, int64_t stride, const int64_t *group_off, const int64_t *group_len, int n_groups,
                                        float grad_scale, void *stream)
{
    return erl_launch_reduce_exchange_f32(grads, 1, stride, grads, group_off, group_len, n_groups, grad_scale, true, nullptr, (hipStream_t)stream);
}

int erl_clip_adam_partials_images_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t stride,
                                      const int64_t *group_off, const int64_t *group_len, int n_groups, int32_t step, float lr, float beta1,
                                      float beta2, float eps, float max_norm, float grad_scale, const S3Images *images, const uint32_t *poison,
                                      void *stream)
{
    ERL_REQUIRE(params && grads && exp_avg && exp_avg_sq && n_groups >= 1 && step >= 1, "erl_clip_adam_partials_f32: bad argument");
    TailGroups gr;
    int rc = fill_groups("erl_clip_adam_partials_f32", group_off, group_len, n_groups, stride, &gr);
    if (rc) return rc;
    const int64_t nblk = erl_cdiv(stride, 64);      // partial norms: one per 64-element chunk and group
    ERL_REQUIRE(nblk <= kMaxPartialChunks, "erl_clip_adam_partials_f32: row too long (%lld floats)", (long long)stride);
    double *partials = nullptr;
    if ((rc = erl_tail_partials(&partials, (hipStream_t)stream))) return rc;
    int64_t longest = 1;
    for (int i = 0; i < n_groups; ++i) longest = gr.len[i] > longest ? gr.len[i] : longest;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(clip_adam_partials_kernel, dim3((unsigned)erl_cdiv(longest, 1024), n_groups), dim3(1024), 0, (hipStream_t)stream, params,
                       grads, exp_avg, exp_avg_sq, gr, partials, (int)nblk, beta1, beta2, eps, max_norm, grad_scale, (float)((double)lr / bc1),
                       (float)sqrt(bc2), images ? *images : S3Images{}, poison, erl_span_slot(ERL_SPAN_CLIP_ADAM, erl_cdiv(longest, 1024) * n_groups));
    ERL_LAUNCH_CHECK("erl_clip_adam_partials_f32");
}

// clip + Adam of ONE parameter group behind erl_launch_reduce_group_f32 on the same stream
int erl_clip_adam_partials_group_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t stride, const int64_t *group_off,
                                     const int64_t *group_len, int n_groups, int group, int32_t step, float lr, float beta1, float beta2, float eps,
                                     float max_norm, float grad_scale, const S3Images *images, void *stream)
{
    ERL_REQUIRE(params && grads && exp_avg && exp_avg_sq && n_groups >= 1 && group >= 0 && group < n_groups && step >= 1,
                "clip + Adam (one group): bad argument");
    TailGroups gr;
    int rc = fill_groups("clip + Adam (one group)", group_off, group_len, n_groups, stride, &gr);
    if (rc) return rc;
    const int64_t nblk = erl_cdiv(stride, 64);
    ERL_REQUIRE(nblk <= kMaxPartialChunks && gr.len[group] >= 1, "clip + Adam (one group): row too long / empty group");
    double *partials = nullptr;
    if ((rc = erl_tail_partials(&partials, (hipStream_t)stream))) return rc;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t nb = erl_cdiv(gr.len[group], 1024);
    hipLaunchKernelGGL(clip_adam_partials_kernel, dim3((unsigned)nb, 1), dim3(1024), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, gr,
                       partials, (int)nblk, beta1, beta2, eps, max_norm, grad_scale, (float)((double)lr / bc1), (float)sqrt(bc2),
                       images ? *images : S3Images{}, (const uint32_t *)nullptr, erl_span_slot(ERL_SPAN_CLIP_ADAM, nb), group, true);
    ERL_LAUNCH_CHECK("clip + Adam (one group)");
}

extern "C" int erl_clip_adam_partials_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t stride,
                                          const int64_t *group_off, const int64_t *group_len, int n_groups, int32_t step, float lr, float beta1,
                                          float beta2, float eps, float max_norm, float grad_scale, void *stream)
{
    return erl_clip_adam_partials_images_f32(params, grads, exp_avg, exp_avg_sq, stride, group_off, group_len, n_groups, step, lr, beta1, beta2,
                                             eps, max_norm, grad_scale, nullptr, nullptr, stream);
}

// the same with a communicator: a peer-to-peer communicator whose exchange timed out in this update loop is poisoned, and the
// update is then SKIPPED (parameters, moments untouched) -- erl_comm_reduce_exchange_f32 + this = the data-parallel tail
extern "C" int erl_comm_clip_adam_partials_f32(void *comm, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t stride,
                                               const int64_t *group_off, const int64_t *group_len, int n_groups, int32_t step, float lr,
                                               float beta1, float beta2, float eps, float max_norm, float grad_scale, void *stream)
{
    return erl_clip_adam_partials_images_f32(params, grads, exp_avg, exp_avg_sq, stride, group_off, group_len, n_groups, step, lr, beta1, beta2,
                                             eps, max_norm, grad_scale, nullptr, erl_comm_poison_word(comm), stream);
}

// Image buffers: library-owned, one pair per (device, stream) that ran an update loop (a handful in any process); rebuilt from
// the fp32 parameters at the start of every loop, so nothing has to be kept coherent across calls.
namespace {
struct S3Slot {
    int device = -1;
    hipStream_t stream = nullptr;
    unsigned char *buf = nullptr;
    size_t bytes = 0;
    float *aux = nullptr;         // per-sample records (s3_image.h), grown on demand
    size_t aux_bytes = 0;
};
S3Slot g_s3_slots[16];
constexpr size_t kS3AuxMaxBytes = (size_t)256 << 20;     // beyond it (H x N > 4 Mi rows) the minibatch kernels keep gathering from the buffers
}  // namespace

int erl_s3_images_build(const float *flat_params, int S, int h1, int h2, int A, S3Images *out, const double *adv_partials, int n_partials,
                        int64_t H, int64_t N, double *adv_stats, hipStream_t stream, const S3AuxSrc *aux_src)
{
    int dev = 0;
    int rc = erl_hip_status(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    const size_t two = (s3_image_bytes(h1, h2) + 1023) / 1024 * 1024, first = (s3_image1_bytes(h1, S) + 1023) / 1024 * 1024;
    const size_t one = two + first, need = 2 * one;
    S3Slot *slot = nullptr;
    for (auto &s : g_s3_slots)
        if (s.buf && s.device == dev && s.stream == stream) slot = &s;
    if (!slot)
        for (auto &s : g_s3_slots)
            if (!s.buf) { slot = &s; break; }
    ERL_REQUIRE(slot, "erl_s3_images_build: more than 16 (device, stream) pairs ran a split-arithmetic update loop");
    if (slot->bytes < need) {
        if (slot->buf) {
            if ((rc = erl_hip_status(hipStreamSynchronize(slot->stream), "hipStreamSynchronize"))) return rc;
            if ((rc = erl_hip_status(hipFree(slot->buf), "hipFree"))) return rc;
            slot->buf = nullptr;
        }
        if ((rc = erl_hip_status(hipMalloc((void **)&slot->buf, need), "hipMalloc(weight images)"))) return rc;
        slot->bytes = need;
    }
    slot->device = dev;
    slot->stream = stream;
    const int64_t Pa = (int64_t)h1 * S + h1 + (int64_t)h2 * h1 + h2 + (int64_t)A * h2 + A + A;     // actor block incl. action_std_log
    for (int gi = 0; gi < 2; ++gi) {
        out->net[gi].img = slot->buf + gi * one;
        out->net[gi].w2_off = (int64_t)h1 * S + h1;
        out->net[gi].h1 = h1;
        out->net[gi].h2 = h2;
        out->net[gi].img1 = slot->buf + gi * one + two;
        out->net[gi].S = S;
        out->net[gi].K1 = s3_image_k1(S);
    }
    const int64_t elems = (int64_t)h1 * h2 + (int64_t)h1 * s3_image_k1(S);
    // the per-sample records ride the same launch (ERL_K6_AUX=0: none -- A/B runs; read per call)
    out->aux = nullptr;
    S3AuxSrc ax{};
    const bool aux_on = [] { const char *e = getenv("ERL_K6_AUX"); return !e || atoi(e) != 0; }();
    if (aux_on && aux_src && aux_src->actions && aux_src->logprobs && aux_src->advantages && aux_src->reward_sums && aux_src->unmasks &&
        aux_src->A >= 1 && aux_src->A <= 8 && aux_src->rows >= 1 && (size_t)aux_src->rows * kS3AuxFloats * sizeof(float) <= kS3AuxMaxBytes) {
        const size_t want = (size_t)aux_src->rows * kS3AuxFloats * sizeof(float);
        if (slot->aux_bytes < want) {
            if (slot->aux) {
                if ((rc = erl_hip_status(hipStreamSynchronize(slot->stream), "hipStreamSynchronize"))) return rc;
                if ((rc = erl_hip_status(hipFree(slot->aux), "hipFree"))) return rc;
                slot->aux = nullptr;
                slot->aux_bytes = 0;
            }
            if ((rc = erl_hip_status(hipMalloc((void **)&slot->aux, want), "hipMalloc(per-sample records)"))) return rc;
            slot->aux_bytes = want;
        }
        ax = *aux_src;
        out->aux = slot->aux;
    }
    const int64_t gx_w = erl_cdiv(elems, 256), gx_a = out->aux ? erl_cdiv(ax.rows, 256) : 0, gx = gx_a > gx_w ? gx_a : gx_w;
    hipLaunchKernelGGL(s3_image_build_kernel, dim3((unsigned)gx, out->aux ? 4 : (adv_partials ? 3 : 2)), dim3(256), 0, stream, flat_params, Pa, *out,
                       adv_partials, n_partials, (int)H, (int)N, adv_stats, ax, slot->aux);
    ERL_LAUNCH_CHECK("erl_s3_images_build");
}
