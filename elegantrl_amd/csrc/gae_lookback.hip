// K3, single-pass time-parallel variant ("LOOKBACK"): 18 B of HBM traffic per (t, n) element -- every input is
// read exactly once, every output written once.  gfx950.
//
// Reference semantics: elegantrl/agents/AgentPPO.py:207-232 (get_advantages) + :146 (reward_sums); the
// recurrence is the affine map  adv_t = delta_t + c_t * adv_{t+1}  documented in gae.hip.
//
// Decomposition.  A workgroup of W waves owns a slab of T = W * L time steps x 256 envs (lane = 4 consecutive
// envs -> one 16-byte load per row and tensor, a wave reads 1 KiB row segments).  Wave w holds its L rows in
// registers (the chunk never goes back to memory between the two local passes):
//   pass 1   per wave: fold the L steps into the chunk's affine map (A, P); stash delta_t in place of r_t
//   LDS      waves exchange their maps; wave w composes the maps of the waves later in time (< w)
//   global   the slab's map is published as ONE 8-byte granule per env {A, tag}; tag = launch nonce << 2 | state.  A
//            slab's P is two-valued (every c_t is 0 or gamma*lambda: P = 0 if the chain is cut inside the slab, else
//            the full-slab product every workgroup can recompute bit-identically), so the state says which:
//            AGG_ZERO / AGG_FULL / PREFIX (inclusive value); a tag with another nonce means "not ready".  The carry
//            into the slab is obtained by walking the later slabs' granules (decoupled look-back).
//   pass 2   per wave: replay the L steps from registers with the true carry, write adv / ret (+ statistics)
// Slabs are handed out through an atomic ticket in launch order, latest time first, so every slab a workgroup
// waits for belongs to a workgroup that is already running (forward progress without co-residency).
// Granules are written/read with 8-byte agent-scope atomics (sc1 write-through stores / L1-bypassing loads):
// self-validating, so no fences are needed (MI355X_MICROARCH.md, handoff-1to1 / R2 granules).
// The granule table and the ticket counter live in a small library-owned device buffer that only this kernel ever
// writes (allocated on first use, one per device): the per-launch nonce makes every older granule read as "not
// ready", so nothing has to be cleared between launches (a memset + its launch gap cost ~25 % at 151 MB).  Problems
// whose table exceeds that buffer, or calls on a second stream, use the caller's workspace with a memset instead.
#include "erl_common.h"

#include <mutex>

namespace {

constexpr int LB_MAX_WAVES = 16;
constexpr uint32_t LB_AGG_ZERO = 1u, LB_AGG_FULL = 2u, LB_PREFIX = 3u;   // granule states (0 never appears with a live nonce)

struct LbArgs {
    float *rewards;
    uint8_t *undones;
    const uint8_t *unmasks;
    const float *values, *next_value;
    float *adv, *ret;
    int H, N, G, K;            // G env groups of 256, K time slabs of T steps
    float gamma, lam;
    int vtrace, mutate;
    uint32_t *ticket;          // monotonically increasing across launches; this launch's tickets start at ticket_base
    uint32_t ticket_base, nonce;
    unsigned long long *slots; // [K][N] granules {A, nonce << 2 | state}
    double *partials;          // [gridDim.x][3]
    uint32_t *fault;           // host-mapped counter of look-back spin timeouts (erl_async_fault_count); may be NULL
    uint32_t spin_limit;       // polls per granule before a predecessor is declared lost
    uint32_t publish_delay;    // ERL_GAE_LB_DELAY=n (tests): every second slab sleeps n x 128 x 64 clocks before it publishes anything
    uint32_t publish_nonce;    // == nonce; ERL_GAE_LB_FAULT=1 (tests) publishes under a foreign nonce so readers time out
    unsigned long long *span;  // measurement hook (erl_common.h: erl_span_*); nullptr = off
};

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long pack_granule(float a, uint32_t tag)
{
    return (unsigned long long)__float_as_uint(a) | ((unsigned long long)tag << 32);
}

template <int L, bool STATS, bool NT>
__global__ __launch_bounds__(L >= 16 ? 512 : LB_MAX_WAVES * 64) void gae_lookback_kernel(LbArgs g)
{
    __shared__ float2 s_agg[LB_MAX_WAVES][256];
    __shared__ float s_carry[256];
    __shared__ double s_red[3][LB_MAX_WAVES];
    __shared__ uint32_t s_ticket;

    const unsigned long long t_span = erl_span_in(g.span);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    if (threadIdx.x == 0) s_ticket = atomicAdd(g.ticket, 1u) - g.ticket_base;   // wrap-safe: tickets of this launch are 0 .. grid-1
    __syncthreads();
    const int ticket = (int)s_ticket;
    const int kk = ticket / g.G, grp = ticket - kk * g.G;             // kk = 0 is the latest slab in time
    const int T = W * L;
    const int n0 = grp * 256 + lane * 4;
    const bool live = n0 < g.N;
    const int t_top = g.H - kk * T - w * L - 1;                       // this wave's latest step; steps t_top - j
    const size_t N = (size_t)g.N;
    const float gl = g.gamma * g.lam;

    // ---- issue every load of the chunk up front (they do not depend on the recurrence)
    float4 r[L], v[L];
    uint32_t ud4[L], um4[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int t = t_top - j;
        if (live && t >= 0) {
            const size_t i = (size_t)t * N + n0;
            if constexpr (NT) {
                // Streaming (non-temporal) loads for the single-use inputs of a LARGE scan (round 6).  The scan starts behind kernels
                // that left the Infinity Cache full of dirty lines (a rollout's buffers, the value pre-pass); ordinary loads allocate
                // there, every allocation evicts a dirty line, and the scan shares HBM with that write-back: 2048 x 4096 behind 640 MB
                // of writes 42.9 us, with streaming loads 29.2 us; in `bench.py --config cd`'s loop 38.9 -> 32.6 us = 0.49 -> 0.58 of
                // the HBM peak.  On inputs that ARE cached (back-to-back calls on the same 84 MB) it costs 2 us (26.3 -> 28.7): the
                // launcher uses it from 32 MB of inputs + outputs on (profiles/r06_gae_lb_sweep.txt, r06_gae_lb_sweep_nt_loads.txt).
                typedef float nt_f4 __attribute__((ext_vector_type(4)));
                const nt_f4 r_ = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(g.rewards + i));
                const nt_f4 v_ = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(g.values + i));
                r[j] = make_float4(r_.x, r_.y, r_.z, r_.w);
                v[j] = make_float4(v_.x, v_.y, v_.z, v_.w);
                ud4[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(g.undones + i));
                um4[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(g.unmasks + i));
            } else {
                r[j] = *reinterpret_cast<const float4 *>(g.rewards + i);
                v[j] = *reinterpret_cast<const float4 *>(g.values + i);
                ud4[j] = *reinterpret_cast<const uint32_t *>(g.undones + i);
                um4[j] = *reinterpret_cast<const uint32_t *>(g.unmasks + i);
            }
        } else {
            r[j] = v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            ud4[j] = 0u;
            um4[j] = 0x01010101u;
        }
    }
    float4 vn = make_float4(0.f, 0.f, 0.f, 0.f);                      // value following the chunk's latest step
    if (live && t_top >= 0) {
        if (t_top + 1 == g.H) { if (g.vtrace) vn = *reinterpret_cast<const float4 *>(g.next_value + n0); }
        else vn = *reinterpret_cast<const float4 *>(g.values + (size_t)(t_top + 1) * N + n0);
    }

    // ---- pass 1: chunk -> affine map (A, P); r[j] <- delta, cmask bit <- "chain continues"
    float A[4] = {0.f, 0.f, 0.f, 0.f}, P[4] = {1.f, 1.f, 1.f, 1.f};
    float vnext[4] = {vn.x, vn.y, vn.z, vn.w};
    uint32_t cmask[(L * 4 + 31) / 32];
#pragma unroll
    for (int q = 0; q < (L * 4 + 31) / 32; ++q) cmask[q] = 0u;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const float ro[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        float dl[4];
        bool fix = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bool ud = ((ud4[j] >> (8 * e)) & 0xFFu) != 0u;
            const bool um = ((um4[j] >> (8 * e)) & 0xFFu) != 0u;
            float rr = ro[e];
            if (!um) {  // truncated: bootstrap with V(s_t) and cut the chain (AgentPPO.py:211-214)
                rr += vv[e];
                ud = false;
                fix = true;
            }
            const float m = ud ? g.gamma : 0.f, c = ud ? gl : 0.f;
            dl[e] = (rr + m * vnext[e]) - vv[e];
            A[e] = dl[e] + c * A[e];
            P[e] = c * P[e];
            vnext[e] = vv[e];
            if (ud) cmask[(j * 4 + e) >> 5] |= 1u << ((j * 4 + e) & 31);
        }
        if (fix && g.mutate) {   // rare (truncations): write the fix-up back like the reference does to its caller
            const size_t i = (size_t)(t_top - j) * N + n0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (((um4[j] >> (8 * e)) & 0xFFu) == 0u) {
                    g.rewards[i + e] = ro[e] + vv[e];
                    g.undones[i + e] = 0;
                }
        }
        r[j] = make_float4(dl[0], dl[1], dl[2], dl[3]);   // pass 2 only needs delta
    }

    // ---- exchange the waves' maps through LDS
    {
        float4 *dst = reinterpret_cast<float4 *>(&s_agg[w][lane * 4]);
        dst[0] = make_float4(A[0], P[0], A[1], P[1]);
        dst[1] = make_float4(A[2], P[2], A[3], P[3]);
    }
    __syncthreads();
    // map from the slab's incoming carry to this wave's incoming carry: compose waves 0 .. w-1 (latest first)
    float cA[4] = {0.f, 0.f, 0.f, 0.f}, cP[4] = {1.f, 1.f, 1.f, 1.f};
    for (int u = 0; u < w; ++u) {
        const float4 *src = reinterpret_cast<const float4 *>(&s_agg[u][lane * 4]);
        const float4 x = src[0], y = src[1];
        const float a_[4] = {x.x, x.z, y.x, y.z}, p_[4] = {x.y, x.w, y.y, y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cA[e] = a_[e] + p_[e] * cA[e];
            cP[e] = p_[e] * cP[e];
        }
    }

    // ---- the earliest wave owns the slab's total map: publish it, look back, publish the inclusive value
    if (w == W - 1) {
        float sA[4], sP[4], carry[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sA[e] = A[e] + P[e] * cA[e];
            sP[e] = P[e] * cP[e];
            carry[e] = 0.f;
        }
        // product of a full slab with no cut, composed exactly like sP above (bit-identical in every workgroup)
        float p_full;
        {
            float pw = 1.f;
#pragma unroll
            for (int j = 0; j < L; ++j) pw = gl * pw;
            float cp = 1.f;
            for (int u = 0; u < W - 1; ++u) cp = pw * cp;
            p_full = pw * cp;
        }
        const uint32_t tagbase = g.publish_nonce << 2;
        unsigned long long *mine = g.slots + (size_t)kk * N + n0;
        const bool has_reader = kk + 1 < g.K;
        if (g.publish_delay && (kk & 1))
            for (uint32_t i = 0; i < g.publish_delay * 128u; ++i) __builtin_amdgcn_s_sleep(127);
        if (live && has_reader) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                __hip_atomic_store(mine + e, pack_granule(sA[e], tagbase | (sP[e] == 0.f ? LB_AGG_ZERO : LB_AGG_FULL)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (live && kk > 0) {
            float accA[4] = {0.f, 0.f, 0.f, 0.f}, accP[4] = {1.f, 1.f, 1.f, 1.f};
            uint32_t open = 0xFu;   // envs whose carry still depends on later slabs
            // The walk over the later slabs' granules, latest first.  Round 5: the lane's four granules of a slab are ONE 32-byte line
            // segment and the granules of several slabs do not depend on each other, so a round fetches LB_BATCH (2 or 4) slabs x 4 envs with
            // two 16-byte agent-scope loads per slab, all in flight together, and consumes them in order while they are valid (round
            // 4 issued one 8-byte load per env and slab and waited for each: up to 4 (K - 1) dependent round trips -- at 200 x 4096,
            // K = 7, the scan's 9.4 us were mostly this chain).  A granule carries its own validity (the launch's nonce), so reading
            // ahead is safe: what is not published yet is simply fetched again.
            constexpr int LB_BATCH = L >= 8 ? 2 : 4;      // (the L = 8 / 16 instantiations sit at their register budgets: 128 / 256)
            int j = kk - 1;
            uint32_t spins = 0;
            while (j >= 0 && open) {
                unsigned long long gr[LB_BATCH][4];
                {
                    // ALL loads of the round and the wait for them in ONE asm statement: the compiler's waitcnt pass does not see
                    // inline-asm loads, so with the wait in a statement of its own nothing kept a register copy (or a spill) from being
                    // scheduled between the loads and the wait and reading a destination before its data had landed (round 5's form;
                    // its wait also sat inside the per-slab loop, which serialised the two slabs' fetches).  Each 16-byte load covers
                    // two granules that their writer publishes with ONE 8-byte store each (`__hip_atomic_store`, a single
                    // global_store_dwordx2): a naturally aligned 16-byte segment of one 32-byte sector is read in one piece, and a
                    // granule is only trusted once its own nonce matches, so a granule is either seen whole or seen stale and fetched
                    // again (stressed by tests/test_kernels_gpu.py::test_gae_lookback_with_delayed_publishers).
                    const unsigned long long *src[LB_BATCH];
#pragma unroll
                    for (int b = 0; b < LB_BATCH; ++b) src[b] = g.slots + (size_t)(j - b >= 0 ? j - b : 0) * N + n0;
                    u64x2 q[2 * LB_BATCH];
                    if constexpr (LB_BATCH == 2) {
                        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                                     "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                                     "global_load_dwordx4 %2, %5, off sc1\n\t"
                                     "global_load_dwordx4 %3, %5, off offset:16 sc1\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
                                     : "v"(src[0]), "v"(src[1])
                                     : "memory");
                    } else {
                        static_assert(LB_BATCH == 2 || LB_BATCH == 4, "the asm fetches two or four slabs");
                        asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                                     "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                                     "global_load_dwordx4 %2, %9, off sc1\n\t"
                                     "global_load_dwordx4 %3, %9, off offset:16 sc1\n\t"
                                     "global_load_dwordx4 %4, %10, off sc1\n\t"
                                     "global_load_dwordx4 %5, %10, off offset:16 sc1\n\t"
                                     "global_load_dwordx4 %6, %11, off sc1\n\t"
                                     "global_load_dwordx4 %7, %11, off offset:16 sc1\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                                     : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3])
                                     : "memory");
                    }
#pragma unroll
                    for (int b = 0; b < LB_BATCH; ++b) {
                        gr[b][0] = q[2 * b].x; gr[b][1] = q[2 * b].y; gr[b][2] = q[2 * b + 1].x; gr[b][3] = q[2 * b + 1].y;
                    }
                }
                int consumed = 0;
#pragma unroll
                for (int b = 0; b < LB_BATCH; ++b) {
                    if (j - b < 0 || !open || consumed != b) continue;
                    bool all_ready = true;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if ((open & (1u << e)) && ((uint32_t)(gr[b][e] >> 34)) != g.nonce) all_ready = false;
                    const bool give_up = !all_ready && b == 0 && spins >= g.spin_limit;
                    if (!all_ready && !give_up) continue;     // slab j - b is not (fully) published yet: fetch again
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!(open & (1u << e))) continue;
                        const bool ready = ((uint32_t)(gr[b][e] >> 34)) == g.nonce;
                        // bounded: a lost predecessor poisons this env's outputs with NaN AND is counted in the host-visible fault word
                        // (erl_async_fault_count -> the caller raises), never a hang
                        if (!ready && g.fault) __hip_atomic_fetch_add(g.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const float ga = ready ? __uint_as_float((uint32_t)gr[b][e]) : __uint_as_float(0x7FC00000u);
                        const uint32_t state = (uint32_t)(gr[b][e] >> 32) & 3u;
                        accA[e] += accP[e] * ga;
                        if (state == LB_AGG_FULL && ready) accP[e] *= p_full;   // the chain runs through slab j - b: keep walking
                        else open &= ~(1u << e);                                 // inclusive value, or an episode boundary cut it
                    }
                    ++consumed;
                }
                j -= consumed;
                if (!consumed) {
                    __builtin_amdgcn_s_sleep(2);
                    ++spins;
                } else {
                    spins = 0;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) carry[e] = accA[e];
        }
        if (live && has_reader) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                __hip_atomic_store(mine + e, pack_granule(sA[e] + sP[e] * carry[e], tagbase | LB_PREFIX), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        *reinterpret_cast<float4 *>(&s_carry[lane * 4]) = make_float4(carry[0], carry[1], carry[2], carry[3]);
    }
    __syncthreads();

    // ---- pass 2: replay from registers with the true carry
    float a[4];
    {
        const float4 c4 = *reinterpret_cast<const float4 *>(&s_carry[lane * 4]);
        const float cin[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = cA[e] + cP[e] * cin[e];
    }
    double s_all = 0, s_sub = 0, q_sub = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int t = t_top - j;
        const float dd[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool cont = (cmask[(j * 4 + e) >> 5] >> ((j * 4 + e) & 31)) & 1u;
            a[e] = dd[e] + (cont ? gl : 0.f) * a[e];
            o[e] = a[e];
        }
        if (live && t >= 0) {
            const size_t i = (size_t)t * N + n0;
            *reinterpret_cast<float4 *>(g.adv + i) = make_float4(o[0], o[1], o[2], o[3]);
            if (g.ret)
                *reinterpret_cast<float4 *>(g.ret + i) = make_float4(o[0] + vv[0], o[1] + vv[1], o[2] + vv[2], o[3] + vv[3]);
            if (STATS) {
                s_all += ((double)o[0] + (double)o[1]) + ((double)o[2] + (double)o[3]);
                if ((t & 3) == 0) {   // n0 is a multiple of 4: env n0 is the only [::4] column of this lane
                    s_sub += o[0];
                    q_sub += (double)o[0] * o[0];
                }
            }
        }
    }
    if (STATS) {
        const double w0 = wave_sum(s_all), w1 = wave_sum(s_sub), w2 = wave_sum(q_sub);
        if (lane == 0) {
            s_red[0][w] = w0;
            s_red[1][w] = w1;
            s_red[2][w] = w2;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double s = 0;
            for (int u = 0; u < W; ++u) s += s_red[threadIdx.x][u];
            g.partials[(size_t)blockIdx.x * 3 + threadIdx.x] = s;
        }
    }
    erl_span_out(g.span, t_span);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// TALL form (round 6): 64 <= H <= 256.  The look-back form cuts such a horizon into slabs of 32 steps and pays ~2 us per publish -> fetch hop
// between workgroups (200 x 4096: 7 slabs, 8.7 us for 14.7 MB); one slab per 256 envs would leave 16 workgroups to move it.  Here ONE
// workgroup owns the whole horizon of 32 envs: 256 threads = 32 time chunks x 8 lanes x 4 envs, a thread holds L = ceil(H / 32) steps of its
// four envs in registers (16-byte loads, 128 contiguous bytes per row and array), folds them into the chunk's affine map, the 32 maps of an
// env meet in LDS (a thread composes the later chunks' maps in time order, at most 31 of them), pass 2 replays from registers.  No tickets,
// no granules, no wait: every byte is read once and all loads of the workgroup are in flight together; N / 32 workgroups (128 at N = 4096).
// Same arithmetic per step and the same composition rule as the look-back kernel (tolerance class: time-parallel, <= 1e-5 max(1, |ref|)).
// ---------------------------------------------------------------------------------------------------------------------------------
// NARROW form (round 6): 64 time chunks x 4 lanes x 4 envs = one workgroup per SIXTEEN envs (rows of 64 contiguous bytes per array), half
// the steps per thread.  Measured (tools/gae_tall_envs_ab.py, profiles/r06_gae_tall_envs_ab.txt): at N = 4096 it does NOT pay although it
// doubles the workgroups to one per CU (200 x 4096: 4.3 us warm / 7.7 cold against 4.4 / 7.2; 128 x 4096: 3.2 / 5.9 against 2.9 / 4.8 --
// the kernel is a fixed ~3 us of launch + one memory round trip + compose + store there, not CU bandwidth); below 128 workgroups it does
// (200 x 2048: 3.5 / 6.1 against 4.4 / 6.6; 200 x 1000: 3.5 / 5.3 against 4.3 / 6.8).  Chosen when N / 32 < 128; ERL_GAE_TALL_ENVS = 32 | 16
// forces.
constexpr int TALL_CHUNKS = 32, TALL_LANES = 8;
template <int L, bool STATS, bool NT, int TCH = 32, int TLN = 8>
__global__ __launch_bounds__(TCH * TLN) void gae_tall_kernel(LbArgs g)
{
    static_assert(TCH * TLN == 256, "one workgroup = 256 threads");
    __shared__ float2 s_agg[TCH][TLN * 4];
    __shared__ double s_red[3][TCH * TLN / 64];

    const unsigned long long t_span = erl_span_in(g.span);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = threadIdx.x / TLN, el = threadIdx.x % TLN;      // chunk 0 is the latest in time
    const int n0 = blockIdx.x * (TLN * 4) + el * 4;
    const bool live = n0 < g.N;
    const int t_top = g.H - c * L - 1;                                 // this thread's latest step; steps t_top - j
    const size_t N = (size_t)g.N;
    const float gl = g.gamma * g.lam;

    float4 r[L], v[L];
    uint32_t ud4[L], um4[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int t = t_top - j;
        if (live && t >= 0) {
            const size_t i = (size_t)t * N + n0;
            if constexpr (NT) {       // streaming loads for a large scan's single-use inputs (see gae_lookback_kernel)
                typedef float nt_f4 __attribute__((ext_vector_type(4)));
                const nt_f4 r_ = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(g.rewards + i));
                const nt_f4 v_ = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(g.values + i));
                r[j] = make_float4(r_.x, r_.y, r_.z, r_.w);
                v[j] = make_float4(v_.x, v_.y, v_.z, v_.w);
                ud4[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(g.undones + i));
                um4[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(g.unmasks + i));
            } else {
                r[j] = *reinterpret_cast<const float4 *>(g.rewards + i);
                v[j] = *reinterpret_cast<const float4 *>(g.values + i);
                ud4[j] = *reinterpret_cast<const uint32_t *>(g.undones + i);
                um4[j] = *reinterpret_cast<const uint32_t *>(g.unmasks + i);
            }
        } else {
            r[j] = v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            ud4[j] = 0u;
            um4[j] = 0x01010101u;
        }
    }
    float4 vn = make_float4(0.f, 0.f, 0.f, 0.f);                      // value following the chunk's latest step
    if (live && t_top >= 0) {
        if (t_top + 1 == g.H) { if (g.vtrace) vn = *reinterpret_cast<const float4 *>(g.next_value + n0); }
        else vn = *reinterpret_cast<const float4 *>(g.values + (size_t)(t_top + 1) * N + n0);
    }

    // ---- pass 1: chunk -> affine map (A, P); r[j] <- delta, cmask bit <- "chain continues"  (gae_lookback_kernel's, step for step)
    float A[4] = {0.f, 0.f, 0.f, 0.f}, P[4] = {1.f, 1.f, 1.f, 1.f};
    float vnext[4] = {vn.x, vn.y, vn.z, vn.w};
    uint32_t cmask[(L * 4 + 31) / 32];
#pragma unroll
    for (int q = 0; q < (L * 4 + 31) / 32; ++q) cmask[q] = 0u;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const float ro[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        float dl[4];
        bool fix = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bool ud = ((ud4[j] >> (8 * e)) & 0xFFu) != 0u;
            const bool um = ((um4[j] >> (8 * e)) & 0xFFu) != 0u;
            float rr = ro[e];
            if (!um) {  // truncated: bootstrap with V(s_t) and cut the chain (AgentPPO.py:211-214)
                rr += vv[e];
                ud = false;
                fix = true;
            }
            const float m = ud ? g.gamma : 0.f, cc = ud ? gl : 0.f;
            dl[e] = (rr + m * vnext[e]) - vv[e];
            A[e] = dl[e] + cc * A[e];
            P[e] = cc * P[e];
            vnext[e] = vv[e];
            if (ud) cmask[(j * 4 + e) >> 5] |= 1u << ((j * 4 + e) & 31);
        }
        if (fix && g.mutate && live && t_top - j >= 0) {   // rare (truncations): write the fix-up back like the reference does to its caller
            const size_t i = (size_t)(t_top - j) * N + n0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (((um4[j] >> (8 * e)) & 0xFFu) == 0u) {
                    g.rewards[i + e] = ro[e] + vv[e];
                    g.undones[i + e] = 0;
                }
        }
        r[j] = make_float4(dl[0], dl[1], dl[2], dl[3]);   // pass 2 only needs delta
    }

    // ---- the chunks' maps meet in LDS; this thread's incoming value = the later chunks 0 .. c-1 composed in time order on a zero carry
    {
        float4 *dst = reinterpret_cast<float4 *>(&s_agg[c][el * 4]);
        dst[0] = make_float4(A[0], P[0], A[1], P[1]);
        dst[1] = make_float4(A[2], P[2], A[3], P[3]);
    }
    __syncthreads();
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < c; ++u) {
        const float4 *src = reinterpret_cast<const float4 *>(&s_agg[u][el * 4]);
        const float4 x = src[0], y = src[1];
        const float a_[4] = {x.x, x.z, y.x, y.z}, p_[4] = {x.y, x.w, y.y, y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = a_[e] + p_[e] * a[e];
    }

    // ---- pass 2: replay from registers
    double s_all = 0, s_sub = 0, q_sub = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int t = t_top - j;
        const float dd[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool cont = (cmask[(j * 4 + e) >> 5] >> ((j * 4 + e) & 31)) & 1u;
            a[e] = dd[e] + (cont ? gl : 0.f) * a[e];
            o[e] = a[e];
        }
        if (live && t >= 0) {
            const size_t i = (size_t)t * N + n0;
            *reinterpret_cast<float4 *>(g.adv + i) = make_float4(o[0], o[1], o[2], o[3]);
            if (g.ret)
                *reinterpret_cast<float4 *>(g.ret + i) = make_float4(o[0] + vv[0], o[1] + vv[1], o[2] + vv[2], o[3] + vv[3]);
            if (STATS) {
                s_all += ((double)o[0] + (double)o[1]) + ((double)o[2] + (double)o[3]);
                if ((t & 3) == 0) {   // n0 is a multiple of 4: env n0 is the only [::4] column of this lane
                    s_sub += o[0];
                    q_sub += (double)o[0] * o[0];
                }
            }
        }
    }
    if (STATS) {
        const double w0 = wave_sum(s_all), w1 = wave_sum(s_sub), w2 = wave_sum(q_sub);
        if (lane == 0) {
            s_red[0][w] = w0;
            s_red[1][w] = w1;
            s_red[2][w] = w2;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double sum = 0;
            for (int u = 0; u < TCH * TLN / 64; ++u) sum += s_red[threadIdx.x][u];
            g.partials[(size_t)blockIdx.x * 3 + threadIdx.x] = sum;
        }
    }
    erl_span_out(g.span, t_span);
}

int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

}  // namespace

// Shape heuristics (overridable for sweeps with ERL_GAE_LB_L in {2,4,8,16} and ERL_GAE_LB_W in 1..16).
void erl_gae_lookback_pick(int64_t H, int64_t N, int *L, int *W)
{
    (void)N;
    int l, w;
    // measured on MI355X (tools/gae_sweep.py, profiles/gae_sweep_r01.json): slabs of 128 steps once H is large,
    // short slabs (more workgroups) for short horizons
    // (L = 16, W = 8: 512 threads, 256 VGPRs, no scratch; the L = 8 instantiation is built for 1024 threads = 64 VGPRs and
    // spills 24-32 bytes per lane -- equally fast at 2048 x 4096 (32.5 vs 33.1 us, profiles/r02_gae_tiling_ab.txt), so the large
    // sizes avoid it; nontemporal output stores were measured there too: no change)
    // round 6 (tools/gae_lb_sweep.py with a cold column, profiles/r06_gae_lb_sweep.txt): 1024 x 4096 runs 16.9 us warm / 23.7 us cold on
    // slabs of 64 steps (L = 4, W = 16: 1024 threads, no spill) against 18.7 / 29.7 us on L = 8 / W = 16 (its instantiation spills)
    if (H >= 2048) { l = 16; w = 8; }
    else if (H >= 512) { l = 4; w = 16; }
    else if (H >= 64) { l = 4; w = 8; }
    else { l = 4; w = (int)((H + 3) / 4); }      // one slab covers the horizon: no look-back at all (round 5: 32 x 32768 18.9 -> 5.5 us)
    l = env_int("ERL_GAE_LB_L", l);
    w = env_int("ERL_GAE_LB_W", w);
    if (l != 2 && l != 4 && l != 8 && l != 16) l = 8;
    if (w < 1) w = 1;
    if (w > LB_MAX_WAVES) w = LB_MAX_WAVES;
    if (l >= 16 && w > 8) w = 8;   // the L = 16 instantiation is compiled for <= 512 threads (256 VGPRs)
    *L = l;
    *W = w;
}

bool erl_gae_lookback_usable(const float *rewards, const uint8_t *undones, const uint8_t *unmasks, const float *values,
                             const float *next_value, const float *adv, const float *ret, int64_t N)
{
    const uintptr_t f = reinterpret_cast<uintptr_t>(rewards) | reinterpret_cast<uintptr_t>(values) |
                        reinterpret_cast<uintptr_t>(next_value) | reinterpret_cast<uintptr_t>(adv) | reinterpret_cast<uintptr_t>(ret);
    const uintptr_t b = reinterpret_cast<uintptr_t>(undones) | reinterpret_cast<uintptr_t>(unmasks);
    return (N % 4 == 0) && (f % 16 == 0) && (b % 4 == 0);
}

// Library-owned look-back tables, one per (device, stream) that has launched the scan: [ticket counter: 256 B][granules].  Only
// gae_lookback_kernel writes them; stale granules carry older nonces, and launches that share a table are ordered by its stream
// (two agents on two streams of one device each get their own: round 4 kept ONE table per device and sent the second stream to
// the memset path).  Allocated (and zeroed) on first use; beyond kLbMaxTables the caller's workspace + a memset serve.
constexpr size_t kLbTableBytes = 256 + ((size_t)8 << 20);   // 8 MiB of granules: K * N <= 1M (e.g. 2048 x 65536 at T = 128)
constexpr int kLbMaxTables = 16;
struct LbTable {
    char *ptr = nullptr;
    uint32_t nonce = 1, ticket_base = 0;
    int dev = -1;
    hipStream_t stream = nullptr;
};
static LbTable g_lb_table[kLbMaxTables];
static int g_lb_tables = 0;
static std::mutex g_lb_mutex;      // table lookup / creation / nonce hand-out (agents on several devices or streams launch from their own threads)

// Enqueues [memset +] kernel.  workspace layout (fallback path): [ticket: 256 B][slots: K*N*8 B][partials: nblk*24 B].
// Returns the number of statistics partials (blocks) through *nparts and their location through *partials.
int erl_gae_lookback_launch(float *rewards, uint8_t *undones, const uint8_t *unmasks, const float *values,
                            const float *next_value, float *adv, float *ret, int64_t H, int64_t N, float gamma, float lam,
                            bool vtrace, bool mutate, bool want_stats, void *workspace, int64_t workspace_bytes,
                            double **partials, int *nparts, hipStream_t stream)
{
    // 64 <= H <= 256: one workgroup per 32 envs holds the whole horizon (gae_tall_kernel; at 512 x 4096 it only draws with the slabs: 13.1 vs
    // 13.0 us, profiles/r06_gae_tall_sweep.txt); ERL_GAE_TALL=0, or a forced look-back tiling, keeps slabs
    if (H >= 64 && H <= 8 * TALL_CHUNKS && env_int("ERL_GAE_TALL", 1) && !getenv("ERL_GAE_LB_L") && !getenv("ERL_GAE_LB_W") &&
        !env_int("ERL_GAE_LB_FAULT", 0) && !env_int("ERL_GAE_LB_DELAY", 0)) {
        // 32 envs per workgroup; 16 below 128 workgroups (measured: see the NARROW form's note)
        int tall_envs = env_int("ERL_GAE_TALL_ENVS", erl_cdiv(N, 32) < 128 ? 16 : 32);
        if (tall_envs != 16) tall_envs = 32;
        const int chunks = tall_envs == 16 ? 64 : TALL_CHUNKS;
        const int64_t nb = erl_cdiv(N, tall_envs);
        ERL_REQUIRE(nb < (1LL << 30), "erl_gae_scan_f32: grid too large");
        ERL_REQUIRE(256 + nb * 24 <= workspace_bytes, "erl_gae_scan_f32: workspace too small for the one-workgroup scan");
        LbArgs g{};
        g.rewards = rewards; g.undones = undones; g.unmasks = unmasks; g.values = values; g.next_value = next_value;
        g.adv = adv; g.ret = ret;
        g.H = (int)H; g.N = (int)N; g.G = (int)nb; g.K = 1;
        g.gamma = gamma; g.lam = lam; g.vtrace = vtrace; g.mutate = mutate;
        g.partials = (double *)((char *)workspace + 256);
        g.span = erl_span_slot(ERL_SPAN_GAE, nb);
        *partials = g.partials;
        *nparts = (int)nb;
        const int Lt = (int)erl_cdiv(H, chunks);
        const dim3 grid((unsigned)nb), block(TALL_CHUNKS * TALL_LANES);
        const bool nt = env_int("ERL_GAE_NT", 18 * H * N >= (32LL << 20) ? 1 : 0) != 0;
#define TALL_LAUNCH(LL, CH, LN)                                                                                          \
    do {                                                                                                                 \
        if (want_stats && nt) hipLaunchKernelGGL((gae_tall_kernel<LL, true, true, CH, LN>), grid, block, 0, stream, g);  \
        else if (want_stats) hipLaunchKernelGGL((gae_tall_kernel<LL, true, false, CH, LN>), grid, block, 0, stream, g);  \
        else if (nt) hipLaunchKernelGGL((gae_tall_kernel<LL, false, true, CH, LN>), grid, block, 0, stream, g);          \
        else hipLaunchKernelGGL((gae_tall_kernel<LL, false, false, CH, LN>), grid, block, 0, stream, g);                 \
    } while (0)
        if (tall_envs == 16) {
            if (Lt <= 2) TALL_LAUNCH(2, 64, 4);
            else TALL_LAUNCH(4, 64, 4);
        } else if (Lt <= 2) TALL_LAUNCH(2, 32, 8);
        else if (Lt <= 4) TALL_LAUNCH(4, 32, 8);
        else TALL_LAUNCH(8, 32, 8);
#undef TALL_LAUNCH
        return erl_hip_status(hipGetLastError(), "gae_tall_kernel launch");
    }
    int L, W;
    erl_gae_lookback_pick(H, N, &L, &W);
    const int64_t T = (int64_t)L * W;
    const int64_t K = erl_cdiv(H, T), G = erl_cdiv(N, 256);
    ERL_REQUIRE(K * G < (1LL << 30), "erl_gae_scan_f32: lookback grid too large");
    const size_t slot_bytes = ((size_t)K * N * 8 + 255) & ~(size_t)255;
    const size_t need = 256 + slot_bytes + (size_t)(K * G) * 24;
    ERL_REQUIRE((int64_t)need <= workspace_bytes, "erl_gae_scan_f32: workspace too small for the lookback scan");
    char *ws = (char *)workspace;
    LbArgs g;
    g.rewards = rewards; g.undones = undones; g.unmasks = unmasks; g.values = values; g.next_value = next_value;
    g.adv = adv; g.ret = ret;
    g.H = (int)H; g.N = (int)N; g.G = (int)G; g.K = (int)K;
    g.gamma = gamma; g.lam = lam; g.vtrace = vtrace; g.mutate = mutate;
    g.partials = (double *)(ws + 256 + slot_bytes);
    g.fault = erl_fault_word(ERL_FAULT_GAE_LOOKBACK);
    g.span = erl_span_slot(ERL_SPAN_GAE, K * G);
    {
        const int lim = env_int("ERL_GAE_LB_SPIN", 1 << 22);
        g.spin_limit = lim > 0 ? (uint32_t)lim : 1u;
    }
    *partials = g.partials;
    *nparts = (int)(K * G);

    // fast path: the library-owned table (no clearing); fallback: the caller's workspace, cleared by a memset
    LbTable *tab = nullptr;
    int dev = -1;
    std::lock_guard<std::mutex> lock(g_lb_mutex);
    if (!getenv("ERL_GAE_LB_NO_TABLE") && hipGetDevice(&dev) == hipSuccess && dev >= 0 && 256 + slot_bytes <= kLbTableBytes) {
        for (int i = 0; i < g_lb_tables && !tab; ++i)
            if (g_lb_table[i].dev == dev && g_lb_table[i].stream == stream) tab = &g_lb_table[i];
        if (!tab && g_lb_tables < kLbMaxTables) {
            void *p = nullptr;
            if (hipMalloc(&p, kLbTableBytes) == hipSuccess) {
                // zeroed ON THE LAUNCH STREAM: torch's side streams are non-blocking, i.e. not ordered behind the legacy stream a plain
                // hipMemset runs on -- the scan could start while its ticket counter and granules were still being cleared (round 6: an
                // intermittent memory fault in tests/test_kernels_gpu.py::test_gae_lookback_tables_are_per_stream; latent since round 5)
                if (hipMemsetAsync(p, 0, kLbTableBytes, stream) == hipSuccess) {
                    LbTable &t = g_lb_table[g_lb_tables++];
                    t.ptr = (char *)p; t.dev = dev; t.stream = stream;
                    tab = &t;
                } else {
                    (void)hipFree(p);
                }
            }
            (void)hipGetLastError();
        }
    }
    if (tab) {
        g.ticket = (uint32_t *)tab->ptr;
        g.slots = (unsigned long long *)(tab->ptr + 256);
        g.ticket_base = tab->ticket_base;
        g.nonce = tab->nonce;
        tab->ticket_base += (uint32_t)(K * G);
        if (++tab->nonce >= (1u << 30)) {   // nonce space exhausted: clear behind this launch and start over
            tab->nonce = 1;
        }
    } else {
        g.ticket = (uint32_t *)ws;
        g.slots = (unsigned long long *)(ws + 256);
        g.ticket_base = 0;
        g.nonce = 1;
        int rc = erl_hip_status(hipMemsetAsync(ws, 0, 256 + (K > 1 ? slot_bytes : 0), stream), "hipMemsetAsync(lookback slots)");
        if (rc) return rc;
    }
    g.publish_delay = (uint32_t)env_int("ERL_GAE_LB_DELAY", 0);
    g.publish_nonce = env_int("ERL_GAE_LB_FAULT", 0) ? (g.nonce ^ 0x15555555u) & 0x3fffffffu : g.nonce;
    const dim3 grid((unsigned)(K * G)), block(W * 64);
    // streaming input loads from 32 MB of traffic on (see the kernel's load loop); ERL_GAE_NT=0 / 1 forces (A/B runs)
    const bool nt = env_int("ERL_GAE_NT", 18 * H * N >= (32LL << 20) ? 1 : 0) != 0;
#define LB_LAUNCH(LL)                                                                                                  \
    do {                                                                                                               \
        if (want_stats && nt) hipLaunchKernelGGL((gae_lookback_kernel<LL, true, true>), grid, block, 0, stream, g);     \
        else if (want_stats) hipLaunchKernelGGL((gae_lookback_kernel<LL, true, false>), grid, block, 0, stream, g);     \
        else if (nt) hipLaunchKernelGGL((gae_lookback_kernel<LL, false, true>), grid, block, 0, stream, g);             \
        else hipLaunchKernelGGL((gae_lookback_kernel<LL, false, false>), grid, block, 0, stream, g);                    \
    } while (0)
    switch (L) {
        case 2: LB_LAUNCH(2); break;
        case 4: LB_LAUNCH(4); break;
        case 8: LB_LAUNCH(8); break;
        default: LB_LAUNCH(16); break;
    }
#undef LB_LAUNCH
    if (tab && tab->nonce == 1 && tab->ticket_base != 0) {   // just wrapped: zero the table behind the launch
        (void)hipMemsetAsync(tab->ptr, 0, kLbTableBytes, stream);
        tab->ticket_base = 0;
    }
    return erl_hip_status(hipGetLastError(), "gae_lookback_kernel launch");
}
