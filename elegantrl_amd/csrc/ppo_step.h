// K6 (one PPO minibatch) -- pieces shared by the two kernels behind erl_ppo_step_f32:
//   ppo_step.hip      8 waves x 16 samples, v_mfma_f32_16x16x4_f32 chain (any S <= 128, h1, h2 <= 128)
//   ppo_step_w4.hip   4 waves x 32 samples, v_mfma_f32_32x32x2_f32 chain, one wave per SIMD with the whole activation
//                     set in the 512-entry register file (S <= 64, net [128,128], A <= 8: BASELINE configs 4 / 5)
// Both write the same gradient slabs (one per 128 samples and network) and the same objective partial sums.
#pragma once
#include <functional>
#include "mlp_chain.h"
#include "ppo_objective.h"

struct Ppo2Args {
    const float *P[2];    // actor, critic flat params
    const float *avg[2];
    const float *sd[2];
    const float *states, *actions, *logprobs, *advantages, *reward_sums;
    const uint8_t *unmasks;
    const int64_t *ids;
    int64_t H, N, B;
    int S, h1, h2, A;
    float ratio_clip, lambda_entropy, inv_batch;
    int objective;        // ERL_PPO_OBJ_* (ppo_objective.h)
    float *slabs;
    int64_t stride, Pa, Pc;
    const double *adv_stats;         // nullptr: `advantages` are normalised already; else the raw sums of erl_gae_scan_f32 / the rollout epilogue
    const unsigned char *w2img[2];   // split-arithmetic kernel: pre-split W2 images (s3_image.h) or nullptr
    const unsigned char *w1img[2];   // ... and W1 images (columns padded to 32 / 64); both or neither
    const float *aux = nullptr;      // per-sample records [H N][16] (s3_image.h S3Images::aux; built by the update loop) or nullptr: gather from the buffers
    unsigned long long *span;   // measurement hook (api.cpp, erl_k6_timing_*): this launch's records, kSpanWords u64 per workgroup (see span_enter); nullptr = off
    const int64_t *next_ids;   // the NEXT minibatch's ids (update loops; nullptr: none / unknown): -DERL_K6_EXP & 4 lets the critic's workgroups,
                               // which finish ~5k cycles before the actor's, pull that minibatch's rows towards their XCD's L2
    int exp_net;          // -DERL_K6_EXP & 16 builds only (diagnostics): 0 = every workgroup runs the ACTOR's code path, 1 = the critic's, else by blockIdx.y
    int wg_map;           // the minibatch kernels' workgroup -> (network, slab) map: 0 = (blockIdx.y, blockIdx.x); 1 = by XCD, 2 = by shader engine (k6_wg_map below)
    // the update loop's FIRST launch on a device whose instruction caches miss slowly (the workgroup-map measurement chose map 2): every
    // workgroup first pulls the kernel's own code into its XCD's L2 with data loads (k6_code_touch below; 0 bytes: off).  pc_out: a one-off
    // launch that only reports the kernel's program counter (how the host learns where the code lives; nullptr otherwise)
    const unsigned char *code_touch = nullptr;
    unsigned code_touch_bytes = 0;
    unsigned long long *pc_out = nullptr;
    // >= 0: a HALF launch, grid (n_slabs, 1): every workgroup works on this network (0 actor, 1 critic) -- the two-chain update loop of
    // comm.cpp runs the two networks' minibatch kernels as independent launches on two streams; -1: both networks, grid (n_slabs, 2)
    int only_net = -1;
    long long *prof;      // ERL_PROFILE builds only: [net][8 waves][32] s_memtime stamps of workgroup prof_block
    int prof_block;
};

// how the gradient slabs leave the kernel (26 MB per launch, read once by the slab reduction).  Default since round 4: write-through
// `sc1` stores (the line is dropped from the XCD's L2 as it is written): 2.26-2.32 ms per config-4 iteration against 2.34-2.42 with
// the non-temporal stores of round 2 on four boxes of the pool -- the minibatch kernel itself is as fast, the slab reduction behind it
// 1.1-1.4 us faster per minibatch (tools/r04_slab_policy.sh, profiles/r04_slab_policy.txt).  ERL_SLAB_ST (A/B builds only):
// 0 non-temporal, 1 plain (kernel -1.5 us, reduction +2.4), 2 `sc1`, 3 `sc0 sc1` (= 2), 4 `sc1 nt` (between 0 and 2)
#ifndef ERL_SLAB_ST
#define ERL_SLAB_ST 2
#endif

namespace {

__device__ __forceinline__ void slab_store(float v, float *p)
{
#if ERL_SLAB_ST == 0
    __builtin_nontemporal_store(v, p);
#elif ERL_SLAB_ST == 1
    *p = v;
#elif ERL_SLAB_ST == 2
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif ERL_SLAB_ST == 3
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
    asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
#endif
}

// the same store under a named policy (A/B builds, -DERL_K6_EXP: the LAST weight-gradient pass by plain stores -- the kernel ends in ~5k
// cycles of write-through drain; what is written last could retire at the L2 and leave with the kernel-end write-back instead)
template <int POLICY>
__device__ __forceinline__ void slab_store_as(float v, float *p)
{
    if (POLICY == 1) *p = v;
    else if (POLICY == 2) slab_store(v, p);         // (diagnostics, -DERL_K6_EXP & 32: the caller guards it with a condition that is never true at run time)
    else slab_store(v, p);
}
#ifndef ERL_K6_EARLY_LOGS
#define ERL_K6_EARLY_LOGS 1      // s3: the logged sums' wave parts formed at the objective, not behind four barriers at the kernel's end (round 6)
#endif
#ifndef ERL_K6_SLICE
#define ERL_K6_SLICE 1           // s3 forward / backward: a k-step's operand split spread over all of its MFMA gaps, independent pairs side by side (round 6)
#endif
#ifndef ERL_K6_EXP
#define ERL_K6_EXP 0
#endif
// order of the weight-gradient phases of ppo_step_s3_kernel: 0 = dW1, dW3, dW2 (rounds 3-5), 1 = dW1, dW2, dW3 (the large store first),
// 2 = dW1, dW2, dW3 with the next phase's operand images staged under the current phase's MFMAs (round 6, default)
#ifndef ERL_K6_DW_ORDER
#define ERL_K6_DW_ORDER 2
#endif

constexpr int PB = 128;        // samples per workgroup
// leading dimension of the staged feature-major tiles T[feature][sample]: 16-byte aligned rows, consecutive rows 16
// bytes apart modulo the 128-byte bank span => 8 consecutive rows form one conflict-free ds_read_b128 wavefront slice,
// and a transposing ds_write_b32 of a D-layout tile lands 2 lanes per bank (the minimum for 64 lanes).
constexpr int PLD = PB + 4;

// (adv - mean(adv)) / (std(adv[::4, ::4]) + 1e-5)  (elegantrl/agents/AgentPPO.py:149) from the five raw sums: the arithmetic of
// adv_normalize_kernel (gae.hip), operation for operation, so that a minibatch kernel that normalises at its row load sees the bits
// the separate launch would have written
struct AdvNorm {
    float mean, denom;
    bool on;
};
// computed once at kernel entry (wave-uniform: kept in scalar registers), while the sample ids are in flight
__device__ __forceinline__ AdvNorm adv_norm_consts(const double *__restrict__ stats)
{
#pragma clang fp contract(off)
    AdvNorm n{0.f, 1.f, stats != nullptr};
    if (stats) {
        const double mean_d = stats[0] / stats[1];
        const double cnt = stats[4];
        double var = (stats[3] - stats[2] * stats[2] / cnt) / (cnt - 1.0);  // unbiased (torch.std default)
        var = var > 0 ? var : 0;
        n.mean = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)mean_d)));
        n.denom = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)sqrt(var) + 1e-5f)));
    }
    return n;
}
__device__ __forceinline__ float adv_normalized(float adv, const AdvNorm &n)
{
#pragma clang fp contract(off)
    return n.on ? (adv - n.mean) / n.denom : adv;
}

// which (network, slab) a workgroup of the (n_slabs, 2) grid works on.  Workgroups go to the 8 XCDs round-robin by their linear id
// (MI355X_MICROARCH.md; round 5's per-workgroup records: XCC_ID == linear id % 8 on every box sampled), so map 0 -- network = blockIdx.y
// -- puts 16 actor and 16 critic workgroups on every XCD, 4 + 4 on every shader array: each instruction cache sees BOTH networks' code
// paths (2 x 55 KB against 64 KB).  On most boxes that costs nothing; on ~1 box in 4 the same launch takes 55 us instead of 36, and on
// those boxes a launch whose workgroups all run ONE network's path runs at the fast boxes' speed (DESIGN.md "K6 in round 5").  Map 1:
// XCDs 0-3 run the actor's workgroups, XCDs 4-7 the critic's -- linear id L: network = (L % 8) / 4, slab = 4 * (L / 8) + L % 4 -- so an
// instruction cache holds one path; the tail of a grid whose slab count is not a multiple of 4 falls back to halves.
struct K6Wg {
    bool actor;
    int slab;
};
__device__ __forceinline__ K6Wg k6_wg_map(const Ppo2Args &g)
{
    if (g.only_net >= 0) return K6Wg{g.only_net == 0, (int)blockIdx.x};
    if (g.wg_map == 0) return K6Wg{blockIdx.y == 0, (int)blockIdx.x};
    // maps 1 and 2: groups of 2 * W consecutive linear ids, the first W of a group to the actor, the rest to the critic, slab = W * group +
    // id % W.  W = 4 (map 1): a group is one round over the 8 XCDs -- XCDs 0-3 the actor's.  W = 16 (map 2): a group is one round over the
    // 8 XCDs x 4 shader engines (engine = (id / 8) % 4 by the same records) -- engines 0-1 of EVERY XCD the actor's, 2-3 the critic's: an
    // instruction cache (shared by neighbouring CUs of one engine) still holds one path, and every XCD keeps 16 + 16 workgroups.
    const int W = g.wg_map == 2 ? 16 : 4;
    const int n = (int)gridDim.x, L = (int)(blockIdx.x + gridDim.x * blockIdx.y), full = n / W * W;
    if (L < 2 * full) return K6Wg{L % (2 * W) < W, W * (L / (2 * W)) + L % W};
    const int r = n - full, l = L - 2 * full;                 // 2 r workgroups left for slabs full .. n - 1
    return K6Wg{l < r, full + (l < r ? l : l - r)};
}

// entry / exit of a workgroup on the device's constant-rate clock (sampled launches only).  Every workgroup leaves ONE record of
// kSpanWords u64 at span + kSpanWords * (blockIdx.x + gridDim.x * blockIdx.y), by plain stores from its thread 0 -- no atomics: round 5's
// first version folded the stamps into one {min, max, sums} slot per launch with 5-13 device-scope atomics per workgroup, and 2000
// same-address atomics cost the kernel 6 us (the slab reduction, 795 workgroups: 12 us); the host folds the records instead (api.cpp).
// Record: [0] entry, [1] exit on the constant-rate clock, [2] entry, [3] exit on the SHADER clock (s_memtime: the ratio of the two
// intervals is the clock the kernel actually ran at -- the chip clocks to its power budget, MI355X_MICROARCH.md "DVFS give-back"),
// [4 .. 4 + kSpanPhases / 2): the low words of the shader clock at the kSpanPhases - 1 phase boundaries of kernels that stamp phases
// (ppo_step_s3_kernel), two per u64, 0 otherwise; [7] where the workgroup ran: HW_REG_HW_ID (cu / sh / se ids) | HW_REG_XCC_ID << 32.
constexpr int kSpanPhases = 7;
constexpr int kSpanWords = 8;
static_assert(4 + (kSpanPhases + 1) / 2 <= kSpanWords, "span record too small");
struct SpanT {
    unsigned long long wall, mem;
};
// phase stamps: wave 0's lane 0 parks the shader clock's low word in LDS (t[k], k >= 1: the end of phase k - 1; a phase is far shorter
// than 2^32 cycles, so differences of low words are exact) -- held in scalar registers from stamp to kernel exit they cost the
// minibatch kernel 20-50 more scalar-register spills (it runs at 104 of them)
struct SpanStamps {
    uint32_t *t;                  // LDS, kSpanPhases words
};
__device__ __forceinline__ unsigned long long erl_memtime()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}
__device__ __forceinline__ SpanT span_enter(const Ppo2Args &g)
{
    SpanT t{0ull, 0ull};                          // wave-uniform: scalar register pairs, no vector registers
    if (g.span) { t.wall = wall_clock64(); t.mem = erl_memtime(); }
    return t;
}
// a phase boundary (sampled launches only: a scalar branch otherwise)
#ifndef ERL_NO_SPAN_STAMPS
#define SPAN_STAMP(st, k) do { if (g.span) { const uint32_t t_ = (uint32_t)erl_memtime(); if (threadIdx.x == 0) (st).t[k] = t_; } } while (0)
#else      // A/B builds: the kernel without its phase stamps (what they cost the unsampled launches)
#define SPAN_STAMP(st, k) do { } while (0)
#endif
__device__ __forceinline__ void span_exit(const Ppo2Args &g, SpanT t0, const SpanStamps *st = nullptr)
{
    if (g.span && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left
        const unsigned long long w1 = wall_clock64(), m1 = erl_memtime();
        unsigned long long *rec = g.span + (size_t)kSpanWords * (blockIdx.x + gridDim.x * (g.only_net >= 0 ? g.only_net : blockIdx.y));
        unsigned long long ph[4] = {0ull, 0ull, 0ull, 0ull};
        if (st) {
#pragma unroll
            for (int k = 1; k < kSpanPhases; ++k) ph[(k - 1) >> 1] |= (unsigned long long)st->t[k] << (32 * ((k - 1) & 1));
        }
        uint32_t hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_id), "=s"(xcc_id));
        rec[0] = t0.wall; rec[1] = w1; rec[2] = t0.mem; rec[3] = m1;
        rec[4] = ph[0]; rec[5] = ph[1]; rec[6] = ph[2]; rec[7] = (unsigned long long)hw_id | ((unsigned long long)xcc_id << 32);
    }
}

#ifdef ERL_PROFILE
#define PROF(i)                                                                                   \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        if (g.prof && (int)blockIdx.x == g.prof_block && lane == 0) g.prof[(net * 8 + wave) * 32 + (i)] = (long long)t_; \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
// the same stamp without draining the vector-memory counter (loads that are meant to stay in flight across it)
#define PROF_NV(i)                                                                                \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        if (g.prof && (int)blockIdx.x == g.prof_block && lane == 0) g.prof[(net * 8 + wave) * 32 + (i)] = (long long)t_; \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
// a stamp taken when the value `v` exists (the wait for its producer -- a load -- falls in front of the stamp; nothing else drained)
#define PROF_X(i, v)                                                                              \
    do {                                                                                          \
        auto pv_ = (v);                                                                           \
        asm volatile("" : "+v"(pv_));                                                             \
        PROF_NV(i);                                                                               \
    } while (0)
#else
#define PROF(i) do { } while (0)
#define PROF_NV(i) do { } while (0)
#define PROF_X(i, v) do { } while (0)
#endif

// dW (nA32*32 x nB32*32) = TA . TB^T over the 128 staged samples; output tiles split over the NW waves.
template <int NW>
__device__ __forceinline__ void weight_grad(const float *TA, int nA32, const float *TB, int nB32, float *__restrict__ dW,
                                            int ldw, int cols_real, int wave, int lane)
{
    const int l31 = lane & 31, hi = lane >> 5;
    const int ntiles = nA32 * nB32;
    for (int tile = wave; tile < ntiles; tile += NW) {
        const int it = tile / nB32, jt = tile - it * nB32;
        const float *a = TA + (32 * it + l31) * PLD + hi;
        const float *b = TB + (32 * jt + l31) * PLD + hi;
        f32x16 acc = {0};
        // the sum over samples is order-free: lane half `hi` takes samples 8 j + 4 hi + {0..3} of every group of 8, so
        // one 16-byte read per operand feeds four MFMAs (k-pair of step s' = samples 8 j + s' and 8 j + 4 + s')
        const float *a4 = a + 3 * hi, *b4 = b + 3 * hi;             // a + hi + 3 hi = row + 4 hi
#pragma unroll 8
        for (int j = 0; j < PB / 8; ++j) {
            const float4 av = *reinterpret_cast<const float4 *>(a4 + 8 * j), bv = *reinterpret_cast<const float4 *>(b4 + 8 * j);
            acc = mfma32(av.x, bv.x, acc);
            acc = mfma32(av.y, bv.y, acc);
            acc = mfma32(av.z, bv.z, acc);
            acc = mfma32(av.w, bv.w, acc);
        }
        const int i = 32 * jt + l31;
        if (i < cols_real) {
#pragma unroll
            for (int r = 0; r < 16; ++r) slab_store(acc[r], dW + (size_t)(32 * it + crow(r, hi)) * ldw + i);
        }
    }
}

// bias gradient: out[f] = sum over the 128 staged samples of T[f][:]; wave w reduces features 16 w .. 16 w + 15 (+ 16 NW ...)
template <int NW>
__device__ __forceinline__ void bias_grad(const float *T, int nfeat, float *__restrict__ out, int wave, int lane)
{
    for (int f0 = 16 * wave; f0 < nfeat; f0 += 16 * NW) {
        const int f = f0 + (lane & 15), p = lane >> 4;
        const float *src = T + f * PLD + 32 * p;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + k);
            s0 += v.x + v.z;
            s1 += v.y + v.w;
        }
        float s = s0 + s1;
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (p == 0 && f < nfeat) out[f] = s;
    }
}

}  // namespace

// ppo_step_w4.hip
bool erl_ppo_w4_supported(int S, int h1, int h2, int A);
int erl_ppo_w4_launch(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream);
// ppo_step_s3.hip
bool erl_ppo_s3_supported(int S, int h1, int h2, int A);
// The first launch of an update loop finds the kernel's code in neither the instruction caches nor the L2s (the rollout ran in between).
// On most boxes that costs 2 us; on the boxes with a slow instruction-cache miss path it costs 60 us (97 us for a 37 us kernel), because
// every cache pulls its ~58 KB line by line from HBM.  With code_touch set, every workgroup starts by reading a slice of the code range as
// DATA (16 bytes per lane; the 32 workgroups of an XCD cover 128 KB): the range arrives in each XCD's L2 in one HBM round trip and the
// instruction fetches that follow hit there.  Returns true when the launch was the one-off program-counter report (the caller returns).
__device__ __forceinline__ bool k6_code_touch(const Ppo2Args &g)
{
    if (g.pc_out) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            unsigned long long pc;
            asm volatile("s_getpc_b64 %0" : "=s"(pc));
            *g.pc_out = pc;
        }
        return true;
    }
    if (g.code_touch_bytes) {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
        const size_t off = ((size_t)(L >> 3) * blockDim.x + threadIdx.x) * 16;          // L % 8 = the XCD, L / 8 = the workgroup's rank on it
        if (off + 16 <= g.code_touch_bytes) {
            const unsigned char *p = g.code_touch + off;
            uint4 v;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        }
    }
    return false;
}
// host side of it (ppo_step.hip): is the NEXT launch on this device the first of an update loop on a slow-fetch device (consumes the
// request), and the code range of a kernel whose program counter is known (the loader's allocation around it, through ROCr)
bool erl_k6_code_touch_wanted(int family);
bool erl_k6_code_range(unsigned long long pc, size_t want_bytes, const unsigned char **base, unsigned *bytes);

// host side of the workgroup map (k6_wg_map above; ppo_step.hip): one decision per device and kernel family (0: the (128 | 64, h2) kernels of
// ppo_step_s3_impl.h, 1: the (256, h2[, h3]) kernels of ppo_step_wd_impl.h); `launch(map)` enqueues the kernel once under a map and
// returns an ERL_* code
int erl_k6_wg_map_for_launch(int family, int n_slabs, hipStream_t st, const std::function<int(int)> &launch);
int erl_ppo_s3_launch(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream);
int erl_ppo_s3_launch_pre(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream);      // ppo_step_s3_pre.hip: W2 images given
