// Fused SAC update step for off-policy batch sizes (config 3: B = 256..1024, net [256,256], 4 critics): the same arithmetic
// as the layered step in sac.hip (AgentSAC.update_objectives, elegantrl/agents/AgentSAC.py:42-86) in 12 launches instead of
// ~52.  At these sizes every dense layer of the layered path is a launch-bound GEMM (~7 us whatever it computes,
// profiles/r02_gemm_small_scaling.txt), so the step's time IS its launch count.  What needs the whole batch -- the weight
// gradients (a reduction over the batch) and the global-norm clip -- stays a launch of its own; everything between two such
// points is ROW-LOCAL and runs as one kernel per 16-sample tile:
//
//   actor_fwd      gather-free read of the sampled rows -> L1 -> L2 -> head (mean, log_std) -> tanh action, log-prob
//   critic_fwd     [state | action] -> shared encoder -> decoder e -> q_e                          grid (tiles, E)
//   critic_train   the same with the loss: q_label from the target's q (min over e), dq, back through the decoder to the
//                  encoder output; leaves the operands of the weight gradients in memory            grid (tiles, E)
//   critic_pg      target ensemble on [state | action_pg]: q_e and dq/d(action)                     grid (tiles, E)
//   actor_bwd      head backward (tanh / reparameterisation / the log-prob quirk) -> L2 -> L1 pre-activation gradients
//   dw_table       ALL weight and bias gradients of a network in one launch (a table of small contractions over the batch)
//   clip + Adam    optim.hip, with the soft target update folded into the critic's (AgentBase.py:270-278)
//
// Inside a tile kernel a layer is computed transposed on v_mfma_f32_16x16x4_f32 (exact fp32), outT (features x 16 samples) =
// W . inT, the 8 waves of the workgroup splitting the OUTPUT feature tiles (a 256 x 256 layer on 16 samples is 2.1 MFLOP = one
// CU for ~8k cycles; it cannot be spread further without a cross-workgroup exchange, i.e. a launch); a wave's weight rows come
// straight from L2 into registers as MFMA A operands, all issued before the first MFMA; activations cross waves through a
// [sample][feature] LDS image.  Layers with <= 16 outputs (policy head, Q value, action gradient) split the REDUCTION over the
// waves instead and meet in LDS in a fixed order.  Everything is deterministic.
//
// Round 4 -- a tile workgroup streams a whole cold 256 x 256 layer through ONE CU (~9 us a layer: the weights were just written by Adam,
// every kernel starts with cold L2s), and wherever the math allows it four workgroups share a tile, each owning 64 of a layer's 256 features
// (`split`, ERL_SAC_SPLIT=0 turns it off):
//   critic_fwd / critic_pg   linear in a slice of the decoder's hidden features (q is a sum over them; the policy-gradient pass's loss gradient is
//                            a constant): `split` partial q / d q/d(action) per decoder, added in slice order by their consumers; no exchange
//   actor_bwd                its heavy layer dH0 = W2^T dZ2 is output-split; no exchange
//   actor_fwd (next state)   the head needs the full second layer: the LAST of a tile's four workgroups to arrive adds the four shares of the
//                            head output in slice order and samples (an arrival counter per tile; nobody waits)
//   critic_train             not split: its backward needs the full q of its own forward, i.e. a wait inside the launch
// The clip + Adam launches sum fp64 squared-norm pieces that dw_table leaves (clip_adam_parts_kernel, optim.hip: no grid-wide wait), and
// ReplayBuffer.sample rides in actor_fwd's prologue when the caller hands over the ring (erl_sac_update_ring_f32): 9 launches on the critical
// path, 142 us per update at config 3 (235 in round 3).  The same tile code, looped over H steps with the actor's weights kept in registers /
// LDS, is the persistent off-policy rollout (sac_rollout_synenv_kernel).
#include "mlpn_common.h"

namespace {

constexpr int FT = 512, FWV = 8;       // threads / waves per tile workgroup
constexpr int TS = 16;                 // samples per tile
constexpr int LDT = 260;               // LDS image row stride (floats): 16-byte aligned rows, consecutive samples 4 banks apart
constexpr int FMAXW = 256;             // widest hidden layer
constexpr int FMAXE = 8;
constexpr int kQxSplit = 4;             // feature slices per decoder in the split critic training pass (= kCritSplit)
constexpr float kLogSqrt2PiF2 = 0.91893853320467274178f;

struct FusedDims {
    int S, A, E, h0, h1;
    int64_t B;
    // actor block: Linear(S, h0) GELU Linear(h0, h1) GELU Linear(h1, 2A)
    int64_t aW1, ab1, aW2, ab2, aWh, abh;
    // critic block: encoder Linear(S + A, h0) | decoder e: Linear(h0, h1) GELU Linear(h1, 1), `dec` floats apart
    int64_t cWe, cbe, cdec0, dW1, db1, dWo, dbo, dec;
};

#ifdef ERL_PROFILE
// profiling builds (make EXTRA=-DERL_PROFILE): cycle stamps of workgroup (0, 0), wave 0, per kernel slot (tools/sac_fused_profile.py)
__device__ long long g_fprof[8][32];
#define FPROF(slot, i)                                                                                       \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                                        \
            unsigned long long t_;                                                                           \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            g_fprof[slot][i] = (long long)t_;                                                                \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#else
#define FPROF(slot, i) do { } while (0)
#endif

struct LaneId {
    int tid, wave, lane, l15, q;
};
// a 16-byte write-through store: visible to the other XCDs once acknowledged (s_waitcnt vmcnt(0)), no fence, no whole-L2 write-back
__device__ __forceinline__ void st4_sc1(float *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ LaneId lane_id()
{
    LaneId L;
    L.tid = threadIdx.x; L.lane = L.tid & 63; L.l15 = L.lane & 15; L.q = L.lane >> 4;
    L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);        // in an SGPR: conditions on it are scalar branches, not exec masks
    return L;
}

// four consecutive weights W[row][k0 .. k0 + 3] of a [nrows][K] row-major matrix, zeros outside.  VEC: K % 4 == 0, so a group of
// four is entirely inside or outside the row -- ONE 16-byte load that only needs 4-byte alignment (the decoder blocks of the
// critic ensemble are an odd number of floats apart; gfx950 global loads take any dword-aligned address)
template <bool VEC>
__device__ __forceinline__ float4 ldw4_raw(const float *__restrict__ W, int row, int nrows, int K, int k0)
{
    // the load alone, from an address clamped into the matrix: the caller issues a whole layer's loads back to back, fences the
    // scheduler, and only then applies ldw4_mask (a select or multiply right behind each load makes hipcc wait for every load
    // before it issues the next: measured 3-10x on these kernels)
    const float *p = W + (size_t)min(row, nrows - 1) * K;
    float4 v;
    if (VEC) {
        __builtin_memcpy(&v, p + max(min(k0, K - 4), 0), 16);
    } else {
        v.x = p[min(k0, K - 1)]; v.y = p[min(k0 + 1, K - 1)]; v.z = p[min(k0 + 2, K - 1)]; v.w = p[min(k0 + 3, K - 1)];
    }
    return v;
}
template <bool VEC>
__device__ __forceinline__ float4 ldw4_mask(float4 v, int row, int nrows, int K, int k0)
{
    const bool r = row < nrows;
    if (VEC) {
        const float ok = (r && k0 < K) ? 1.f : 0.f;             // K % 4 == 0: the four are inside or outside together
        v.x *= ok; v.y *= ok; v.z *= ok; v.w *= ok;
    } else {
        v.x *= (r && k0 < K) ? 1.f : 0.f; v.y *= (r && k0 + 1 < K) ? 1.f : 0.f;
        v.z *= (r && k0 + 2 < K) ? 1.f : 0.f; v.w *= (r && k0 + 3 < K) ? 1.f : 0.f;
    }
    return v;
}

// width classes a tile kernel is compiled for: hidden width <= 64 / <= 128 / <= 256 -> k-tiles of 16 (as an input), output tiles
// per wave (as an output)
template <int C> struct WClass { static constexpr int KT = C == 0 ? 4 : (C == 1 ? 8 : 16), NU = C == 2 ? 2 : 1; };

// ---------------------------------------------------------------------------------------------------------
// forward layer, output-split: wave w computes the 16-feature output tiles w and w + 8 of  Z^T = W . in^T + b  for the
// workgroup's 16 samples (in: LDS image Tin[sample][feature], K columns, zero padded to a multiple of 16).  Lane
// (l15 = sample, q) ends up with features 16 ot + 4 q + r (r = 0..3) of its sample: z[u][r] for tile u.
// ---------------------------------------------------------------------------------------------------------
template <int KT_, int NU>
struct FwdW {
    float4 v[NU][KT_];
};

// the weights of a layer, issued as early as the kernel knows which layer comes (they depend on nothing): by the time the
// layer's input image is ready they have landed
template <int KT_, int NU, bool VEC>
__device__ __forceinline__ void layer_fwd_load(const float *__restrict__ W, int K, int N, const LaneId &L, FwdW<KT_, NU> &w)
{
    if (VEC) {
        // hidden-width inputs (K a multiple of 16): ONE per-lane base per tile and wave-uniform k-tile offsets (a clamped address
        // per load costs two VGPRs each before the loads issue: 128 loads' worth spills)
        const int KTr = K >> 4;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float *base = W + (size_t)min(16 * (L.wave + FWV * u) + L.l15, N - 1) * K + 4 * L.q;
#pragma unroll
            for (int kt = 0; kt < KT_; ++kt) __builtin_memcpy(&w.v[u][kt], base + 16 * min(kt, KTr - 1), 16);
        }
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int kt = 0; kt < KT_; ++kt) w.v[u][kt] = ldw4_raw<false>(W, 16 * (L.wave + FWV * u) + L.l15, N, K, 16 * kt + 4 * L.q);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int KT_, int NU, bool VEC>
__device__ __forceinline__ void layer_fwd_mma(FwdW<KT_, NU> &w, const float *__restrict__ bias, int K, int N, const float *Tin, const LaneId &L,
                                              f32x4 (&z)[2])
{
    // Hidden-width inputs (VEC: K a multiple of 16): no per-element masks -- 512 multiplies per lane and layer would cost as much
    // as the layer's MFMAs (fp32 MFMA and VALU share the pipe); k-tiles beyond K are skipped by a wave-uniform branch, output tiles
    // beyond N are computed on clamped rows and dropped by the epilogue.  Input layers (K = S or S + A, any value): the partial
    // k-tile is masked (8 loads).
    const int KTr = (K + 15) >> 4;
    if (!VEC) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int kt = 0; kt < KT_; ++kt) w.v[u][kt] = ldw4_mask<false>(w.v[u][kt], 16 * (L.wave + FWV * u) + L.l15, N, K, 16 * kt + 4 * L.q);
        }
    }
    z[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    z[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *xr = Tin + L.l15 * LDT + 4 * L.q;
#pragma unroll
    for (int kt = 0; kt < KT_; ++kt) {
        if (kt < KTr) {                                             // (SGPR condition: a scalar branch)
            const float4 x = *reinterpret_cast<const float4 *>(xr + 16 * kt);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                z[u] = mfma16(w.v[u][kt].x, x.x, z[u]);
                z[u] = mfma16(w.v[u][kt].y, x.y, z[u]);
                z[u] = mfma16(w.v[u][kt].z, x.z, z[u]);
                z[u] = mfma16(w.v[u][kt].w, x.w, z[u]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int f = 16 * (L.wave + FWV * u) + 4 * L.q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float b = bias[min(f + r, N - 1)];
            z[u][r] += (f + r < N) ? b : 0.f;
        }
    }
}

// the epilogue of a hidden layer: optional exact-erf GELU, the LDS image for the next layer, optional copies in memory
// (H: what the next layer's weight gradient contracts with; G: GELU' for a backward pass in another kernel)
// (`ldg`: the rows of gH / gG are ldg floats apart -- a feature slice stored into the full-width matrix; 0: ldg = N)
__device__ __forceinline__ void emit_hidden(const f32x4 (&z)[2], int N, bool gelu, float *Tout, f32x4 (&gk)[2], float *gH, float *gG, int64_t row,
                                            bool valid, const LaneId &L, int ldg = 0)
{
    if (ldg == 0) ldg = N;
    const int NT = (N + 15) >> 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ot = L.wave + FWV * u;
        if (ot >= NT) continue;
        const int f = 16 * ot + 4 * L.q;
        float h[4], g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (gelu) gelu_and_grad_fast(z[u][r], h[r], g[r]);
            else { h[r] = z[u][r]; g[r] = 1.f; }
            if (f + r >= N) { h[r] = 0.f; g[r] = 0.f; }
            gk[u][r] = g[r];
        }
        *reinterpret_cast<float4 *>(Tout + L.l15 * LDT + f) = make_float4(h[0], h[1], h[2], h[3]);
        if (valid && f < N) {                                     // hidden widths are multiples of 16: whole float4s
            if (gH) *reinterpret_cast<float4 *>(gH + row * ldg + f) = make_float4(h[0], h[1], h[2], h[3]);
            if (gG) *reinterpret_cast<float4 *>(gG + row * ldg + f) = make_float4(g[0], g[1], g[2], g[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward through a layer's input, output-split:  dX^T (Kin features x 16 samples) = W^T . dZ^T, W = [Kz][Kin] row-major in
// memory, dZ: LDS image Tin[sample][feature] (Kz columns, zero padded).  Wave w computes input-feature tiles w and w + 8; lane
// (l15, q) ends up with dX features 16 it + 4 q + r of its sample.  The A operand is W^T: lane (i = l15, q) reads
// W[16 kt + 4 q + j][16 it + i] -- 64-byte runs of a row per quarter wave.
// ---------------------------------------------------------------------------------------------------------
template <int KT_, int NU>
struct BwdW {
    float v[NU][KT_][4];
};

// (`ld`: the rows of W are ld floats apart and Kin of their columns, from W on, are used -- a slice of the input features; 0: ld = Kin)
template <int KT_, int NU>
__device__ __forceinline__ void layer_bwd_load(const float *__restrict__ W, int Kz, int Kin, const LaneId &L, BwdW<KT_, NU> &w, int ld = 0)
{
    if (ld == 0) ld = Kin;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int colc = min(16 * (L.wave + FWV * u) + L.l15, Kin - 1);
        if (KT_ == 1) {                                             // Kz <= 16, any value (the policy head's 2 A rows)
#pragma unroll
            for (int j = 0; j < 4; ++j) w.v[u][0][j] = W[(size_t)min(4 * L.q + j, Kz - 1) * ld + colc];
        } else {                                                    // Kz a multiple of 16: one per-lane offset, wave-uniform row offsets
            const int KTr = Kz >> 4;
            const uint32_t voff = (uint32_t)(4 * L.q) * (uint32_t)ld + (uint32_t)colc;
#pragma unroll
            for (int kt = 0; kt < KT_; ++kt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w.v[u][kt][j] = (W + (size_t)(16 * min(kt, KTr - 1) + j) * ld)[voff];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int KT_, int NU>
__device__ __forceinline__ void layer_bwd_mma(BwdW<KT_, NU> &w, int Kz, int Kin, const float *Tin, const LaneId &L, f32x4 (&dx)[2])
{
    const int KTr = (Kz + 15) >> 4;
    if (KT_ == 1) {                                                 // Kz <= 16, any value: rows beyond Kz masked (4 loads per tile)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w.v[u][0][j] *= (4 * L.q + j < Kz) ? 1.f : 0.f;
        }
    }
    dx[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    dx[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *zr = Tin + L.l15 * LDT + 4 * L.q;
#pragma unroll
    for (int kt = 0; kt < KT_; ++kt) {
        if (kt < KTr) {                                             // Kz a multiple of 16 otherwise: whole k-tiles, no masks (see layer_fwd_mma)
            const float4 d = *reinterpret_cast<const float4 *>(zr + 16 * kt);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                dx[u] = mfma16(w.v[u][kt][0], d.x, dx[u]);
                dx[u] = mfma16(w.v[u][kt][1], d.y, dx[u]);
                dx[u] = mfma16(w.v[u][kt][2], d.z, dx[u]);
                dx[u] = mfma16(w.v[u][kt][3], d.w, dx[u]);
            }
        }
    }
}

struct SmallW {
    float v[2][4];
};

template <bool TRANSPOSED, bool VEC>
__device__ __forceinline__ void layer_small_load(const float *__restrict__ W, int K, int N, int ldw, int col0, const LaneId &L, SmallW &w)
{
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                   // K <= 256: k-tiles wave and wave + 8 (zero weights beyond K)
        const int kt = L.wave + FWV * u;
        if (!TRANSPOSED) {         // rows `ldw` floats apart (0: K), K columns of each used
            const float4 v = ldw4_raw<VEC>(W + (size_t)min(L.l15, N - 1) * (size_t)(ldw ? ldw : K), 0, 1, K, 16 * kt + 4 * L.q);
            w.v[u][0] = v.x; w.v[u][1] = v.y; w.v[u][2] = v.z; w.v[u][3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) w.v[u][j] = W[(size_t)min(16 * kt + 4 * L.q + j, K - 1) * ldw + col0 + min(L.l15, N - 1)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <bool TRANSPOSED, bool VEC>
__device__ __forceinline__ void layer_small_mma(SmallW &w, const float *__restrict__ bias, int K, int N, const float *Tin, float *part, float *Yl,
                                                const LaneId &L)
{
    // K is a hidden width (a multiple of 16): whole k-tiles, skipped by a wave-uniform branch beyond K; outputs f >= N are
    // computed on clamped rows / columns and dropped below
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int KTr = (K + 15) >> 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int kt = L.wave + FWV * u;
        if (kt < KTr) {
            const float4 x = *reinterpret_cast<const float4 *>(Tin + L.l15 * LDT + 16 * kt + 4 * L.q);
            acc = mfma16(w.v[u][0], x.x, acc);
            acc = mfma16(w.v[u][1], x.y, acc);
            acc = mfma16(w.v[u][2], x.z, acc);
            acc = mfma16(w.v[u][3], x.w, acc);
        }
    }
    *reinterpret_cast<float4 *>(part + (L.wave * TS + L.l15) * 16 + 4 * L.q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    lds_barrier();
    if (L.tid < TS * 16) {
        const int s = L.tid >> 4, f = L.tid & 15;
        float y = 0.f;
#pragma unroll
        for (int wv = 0; wv < FWV; ++wv) y += part[(wv * TS + s) * 16 + f];
        Yl[s * 16 + f] = f < N ? y + (bias ? bias[min(f, N - 1)] : 0.f) : 0.f;
    }
    lds_barrier();
}

// rows [row0, row0 + 16) of up to two row-major matrices side by side ([X0 (c0 columns) | X1 (c1 columns)]) into an LDS image,
// zero padded to a multiple of 16 columns and beyond the batch; optionally the concatenated rows go to memory as well
__device__ __forceinline__ void load_rows(const float *__restrict__ X0, int c0, const float *__restrict__ X1, int c1, int64_t row0, int64_t B,
                                          float *T, float *gcat, const LaneId &L)
{
    const int C = c0 + c1, CP = ((C + 15) >> 4) << 4;
    for (int e = L.tid; e < TS * CP; e += FT) {
        const int s = e / CP, c = e - s * CP;
        const int64_t row = row0 + s;
        float v = 0.f;
        if (row < B && c < C) v = c < c0 ? X0[row * c0 + c] : X1[row * c1 + (c - c0)];
        T[s * LDT + c] = v;
        if (gcat && row < B && c < C) gcat[row * C + c] = v;
    }
}

// Barriers in the tile kernels order LDS traffic only (lds_barrier: s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would also
// drain the vector-memory counter, i.e. wait for the prefetched weights (a CU streams a cold 256 KB layer at ~30 GB/s: ~9 us) and for
// every global store of the epilogues at each barrier.
// the images must hold finite values everywhere a (zero-weighted) padded k-tile may read
__device__ __forceinline__ void clear_images(float *T0, float *T1, const LaneId &L)
{
    for (int e = L.tid; e < TS * LDT; e += FT) { T0[e] = 0.f; T1[e] = 0.f; }
}

struct TileLds {
    float T0[TS * LDT], T1[TS * LDT];
    float part[FWV * TS * 16];
    float Yl[TS * 16];
    float red[FWV];
};

// ---------------------------------------------------------------------------------------------------------
// actor forward + policy head:  ActorSAC.get_action_logprob (AgentSAC.py:187-199)
// ---------------------------------------------------------------------------------------------------------
struct ActorFwdArgs {
    const float *P;
    FusedDims d;
    const float *X;                // (B, S) sampled state rows
    float *Xcopy;                  // not NULL: the rows are also written here (the off-policy rollout's states[t] = state)
    // Feature split of the second layer (a call that keeps nothing for a backward pass): grid.y = split, workgroup (tile, s) owns hidden
    // features [s d.h1, (s + 1) d.h1) -- d.h1 is the slice's width, h1_full the layer's: 64 KB of W2 instead of 256 KB through one CU -- and
    // leaves its share of the head's output (a sum over the slice: bias from slice 0) in Ypart[s][b][2 A].  The workgroups of a tile count
    // their arrivals in arrive[tile]; the LAST one adds the shares in slice order, samples, and re-arms the counter: nobody waits.
    int split, h1_full;
    float *Ypart;
    unsigned *arrive;
    // OWNER form of the meeting (round 6, last session; yx != NULL): every slice publishes its share as 8-byte {share, nonce} granules in
    // yx[split][B][2 A] (library-owned, never cleared: the per-launch nonce invalidates older contents) and LEAVES -- no wait for the stores'
    // acknowledgement, no arrival counter; the tile's LAST slice in dispatch order (by = split - 1: its partners were dispatched before it
    // and wait for nothing, so it cannot wait for a workgroup that is not running or finished) polls the others' granules, adds the
    // shares in slice order and samples.  One round trip after the slowest slice instead of three (acknowledge, count, fetch).  Bounded
    // spin; a share that never arrives poisons the tile's actions with NaN and is counted in the host-visible fault word.
    unsigned long long *yx;
    uint32_t nonce, spin_limit;
    uint32_t *fault;
    // ReplayBuffer.sample inside this launch (elegantrl/train/replay_buffer.py:120-134; the gather of csrc/gather.hip replay_sample_kernel):
    // rg.ids != NULL makes X the ring's NEXT-state rows of the drawn transitions, read in place, and the tile's slice-0 workgroup also copies
    // the batch out for the launches behind it (state, action, reward, undone, unmask, next_state, ids0, ids1) -- under the weight loads
    // this kernel waits for anyway
    ErlRingSample rg;
    int rg_self;                   // 1: X = the ring's STATE rows of the drawn transitions themselves (not their next states), nothing copied out:
                                   // the policy-gradient sample's forward pass, which then does not depend on launch (1)'s staging (round 6)
    float *o_state, *o_action, *o_reward, *o_undone, *o_unmask, *o_next;
    const float *noise;            // (B, A) or NULL: Philox keyed by (seed, counter, row, a)
    uint64_t seed, counter;
    float *act_t, *lp;             // (B, A), (B,)
    float *eps_out, *Y;            // keep: (B, A) draws used, (B, 2A) raw head output
    float *H0, *G0, *H1, *G1;      // keep: activations / GELU' for the backward pass and the weight gradients
    const float *alpha_log;        // first call of a step: the temperature BEFORE its update is parked in alpha0 (q_label uses it)
    float *alpha0;
};

template <int C0, int C1>
__device__ __forceinline__ void actor_fwd_body(const ActorFwdArgs &g, TileLds &lds, const int bx, const int by)
{
    const LaneId L = lane_id();
    const FusedDims &d = g.d;
    const int64_t row0 = (int64_t)bx * TS, row = row0 + L.l15;
    const bool valid = row < d.B;
    const int64_t fo = (int64_t)by * d.h1;                   // first second-layer feature of this workgroup's slice (0 unsplit)
    const int h1f = g.split > 1 ? g.h1_full : d.h1;
    if (g.alpha0 && bx == 0 && by == 0 && L.tid == 0) g.alpha0[0] = g.alpha_log[0];
    FPROF(0, 0);
    // every weight this wave will use, requested before anything else
    FwdW<4, WClass<C0>::NU> w1;
    FwdW<WClass<C0>::KT, WClass<C1>::NU> w2;
    SmallW wh;
    layer_fwd_load<4, WClass<C0>::NU, false>(g.P + d.aW1, d.S, d.h0, L, w1);
    layer_fwd_load<WClass<C0>::KT, WClass<C1>::NU, true>(g.P + d.aW2 + fo * d.h0, d.h0, d.h1, L, w2);
    layer_small_load<false, true>(g.P + d.aWh + fo, d.h1, 2 * d.A, h1f, 0, L, wh);
    clear_images(lds.T0, lds.T1, L);
    __shared__ int64_t s_row[TS];
    if (g.rg.ids && L.tid < TS) {                          // ids0 = ids % L, ids1 = ids // L  (replay_buffer.py:124-125)
        const int64_t b = row0 + L.tid, id = g.rg.ids[min(b, d.B - 1)];
        int64_t n, t;                                       // (a 64-bit divide is ~10x the instructions of a 32-bit one, on the tile's critical path)
        if ((uint64_t)id <= 0x7fffffffull && (uint64_t)g.rg.sample_len <= 0x7fffffffull) {
            const uint32_t i32 = (uint32_t)id, l32 = (uint32_t)g.rg.sample_len, n32 = i32 / l32;
            n = n32; t = i32 - n32 * l32;
        } else {
            n = id / g.rg.sample_len; t = id - n * g.rg.sample_len;
        }
        s_row[L.tid] = g.rg.row_floats ? n * g.rg.max_size + t : t * g.rg.num_seqs + n;     // (interleaved ring: sequence-major rows)
        if (by == 0 && b < d.B && !g.rg_self) {
            if (g.rg.out_ids0) g.rg.out_ids0[b] = t;
            if (g.rg.out_ids1) g.rg.out_ids1[b] = n;
        }
    }
    lds_barrier();
    FPROF(0, 1);
    if (g.rg.ids) {
        const int S = d.S, A = d.A, CP = ((S + 15) >> 4) << 4;
        for (int e = L.tid; e < TS * CP; e += FT) {        // the next-state rows: the first layer's input image (+ the batch's copy)
            const int s_ = e / CP, c = e - s_ * CP;
            const int64_t b = row0 + s_;
            float v = 0.f;
            if (b < d.B && c < S) {
                const int64_t nx = g.rg_self ? 0 : 1;           // the next state: the following row of the same sequence
                v = g.rg.row_floats ? g.rg.buf_states[(s_row[s_] + nx) * g.rg.row_floats + c] : g.rg.buf_states[(s_row[s_] + nx * g.rg.num_seqs) * S + c];
                if (by == 0 && !g.rg_self) g.o_next[b * S + c] = v;
            }
            lds.T0[s_ * LDT + c] = v;
        }
        if (by == 0 && !g.rg_self) {
            const int W = S + A + 3;
            for (int e = L.tid; e < TS * W; e += FT) {
                const int s_ = e / W, c = e - s_ * W;
                const int64_t b = row0 + s_, rr = s_row[s_];
                if (b >= d.B) continue;
                if (g.rg.row_floats) {                     // one contiguous row [state | action | reward | undone | unmask]
                    const float v = g.rg.buf_states[rr * g.rg.row_floats + c];
                    if (c < S) g.o_state[b * S + c] = v;
                    else if (c < S + A) g.o_action[b * A + (c - S)] = v;
                    else if (c == S + A) g.o_reward[b] = v;
                    else if (c == S + A + 1) g.o_undone[b] = v;
                    else g.o_unmask[b] = v;
                    continue;
                }
                if (c < S) g.o_state[b * S + c] = g.rg.buf_states[rr * S + c];
                else if (c < S + A) g.o_action[b * A + (c - S)] = g.rg.buf_actions[rr * A + (c - S)];
                else if (c == S + A) g.o_reward[b] = g.rg.buf_rewards[rr];
                else if (c == S + A + 1) g.o_undone[b] = g.rg.buf_undones[rr];
                else g.o_unmask[b] = g.rg.buf_unmasks[rr];
            }
        }
    } else {
        load_rows(g.X, d.S, nullptr, 0, row0, d.B, lds.T0, by == 0 ? g.Xcopy : nullptr, L);
    }
    lds_barrier();
    FPROF(0, 2);
    f32x4 z[2], gk[2];
    layer_fwd_mma<4, WClass<C0>::NU, false>(w1, g.P + d.ab1, d.S, d.h0, lds.T0, L, z);
    FPROF(0, 3);
    emit_hidden(z, d.h0, true, lds.T1, gk, by == 0 ? g.H0 : nullptr, by == 0 ? g.G0 : nullptr, row, valid, L);     // (every slice forms the whole first layer; slice 0 keeps it)
    lds_barrier();
    FPROF(0, 4);
    layer_fwd_mma<WClass<C0>::KT, WClass<C1>::NU, true>(w2, g.P + d.ab2 + fo, d.h0, d.h1, lds.T1, L, z);
    FPROF(0, 5);
    emit_hidden(z, d.h1, true, lds.T0, gk, g.H1 ? g.H1 + fo : nullptr, g.G1 ? g.G1 + fo : nullptr, row, valid, L, h1f);   // (a slice keeps its columns of the full-width matrices)
    lds_barrier();
    FPROF(0, 6);
    layer_small_mma<false, true>(wh, fo == 0 ? g.P + d.abh : nullptr, d.h1, 2 * d.A, lds.T0, lds.part, lds.Yl, L);
    FPROF(0, 7);
    if (g.split > 1) {
        // this slice's share of the head output goes to memory past the (per-XCD) L2; the last of the tile's workgroups to arrive adds the
        // shares in slice order -- the sum every order of arrival gives -- into Yl and carries on as the unsplit kernel does
        __shared__ int s_last;
        const int A2 = 2 * d.A;
        if (g.yx) {
            static_assert(kQxSplit == 4, "four shares are polled at once");
            const int s_ = L.tid >> 4, f = L.tid & 15;
            const bool mine = L.tid < TS * 16 && f < A2 && row0 + s_ < d.B;
            const int owner = g.split - 1;
            if (by != owner) {
                if (mine)
                    __hip_atomic_store(g.yx + ((size_t)by * d.B + row0 + s_) * A2 + f,
                                       (unsigned long long)__float_as_uint(lds.Yl[s_ * 16 + f]) | ((unsigned long long)g.nonce << 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (L.tid < TS * 16) {
                float y = 0.f;
                if (mine) {
                    unsigned long long gr4[kQxSplit];
#pragma unroll
                    for (int k = 0; k < kQxSplit; ++k)
                        gr4[k] = __hip_atomic_load(g.yx + ((size_t)min(k, max(owner - 1, 0)) * d.B + row0 + s_) * A2 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int k = 0; k < kQxSplit; ++k) {
                        if (k >= g.split) continue;
                        float sh;
                        if (k == owner) {
                            sh = lds.Yl[s_ * 16 + f];                   // (my own share: the bits the granule would carry)
                        } else {
                            const unsigned long long *src = g.yx + ((size_t)k * d.B + row0 + s_) * A2 + f;
                            unsigned long long gr = gr4[k];
                            for (uint32_t spins = 0; (uint32_t)(gr >> 32) != g.nonce && spins < g.spin_limit; ++spins) {
                                __builtin_amdgcn_s_sleep(1);
                                gr = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            const bool ready = (uint32_t)(gr >> 32) == g.nonce;
                            if (!ready && g.fault) __hip_atomic_fetch_add(g.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            sh = ready ? __uint_as_float((uint32_t)gr) : __uint_as_float(0x7FC00000u);
                        }
                        y = k == 0 ? sh : y + sh;                        // slice order: ((s0 + s1) + s2) + s3
                    }
                }
                lds.Yl[s_ * 16 + f] = y;
            }
            lds_barrier();
        } else {
        if (L.tid < TS * 16) {
            const int s_ = L.tid >> 4, f = L.tid & 15;
            if (f < A2 && row0 + s_ < d.B)
                __hip_atomic_store(g.Ypart + ((size_t)by * d.B + row0 + s_) * A2 + f, lds.Yl[s_ * 16 + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (L.tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(g.arrive + bx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)g.split - 1u;
            if (s_last) __hip_atomic_store(g.arrive + bx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-armed for the next launch
        }
        __syncthreads();
        if (!s_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (L.tid < TS * 16) {
            const int s_ = L.tid >> 4, f = L.tid & 15;
            float y = 0.f;
            if (f < A2 && row0 + s_ < d.B) {
                // (all shares requested together -- one round trip instead of one per slice -- and added in slice order: the same bits)
                static_assert(kQxSplit == 4, "four shares are fetched at once (kCritSplit below is the same 4)");
                float yk[kQxSplit];
#pragma unroll
                for (int k = 0; k < kQxSplit; ++k)
                    yk[k] = __hip_atomic_load(g.Ypart + ((size_t)min(k, g.split - 1) * d.B + row0 + s_) * A2 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                y = yk[0];
#pragma unroll
                for (int k = 1; k < kQxSplit; ++k)
                    if (k < g.split) y += yk[k];
            }
            lds.Yl[s_ * 16 + f] = y;
        }
        lds_barrier();
        }   // (last-arriver form)
    }
    if (L.tid < TS) {
        const int64_t b = row0 + L.tid;
        if (b < d.B) {
            const int A = d.A;
            float lp = 0.f;
            for (int a = 0; a < A; ++a) {
                const float mean = lds.Yl[L.tid * 16 + a], ls = lds.Yl[L.tid * 16 + A + a];
                const float lsc = fminf(fmaxf(ls, -16.f), 2.f);
                const float sd = expf(lsc);
                const float eps = g.noise ? g.noise[b * A + a] : philox_normal(g.seed, g.counter, (uint32_t)b, (uint32_t)a);
                const float t = tanhf(mean + sd * eps);
                g.act_t[b * A + a] = t;
                if (g.eps_out) g.eps_out[b * A + a] = eps;
                if (g.Y) { g.Y[b * 2 * A + a] = mean; g.Y[b * 2 * A + A + a] = ls; }
                lp += (-logf(sd) - kLogSqrt2PiF2) - logf(-(t * t) + 1.000001f);
            }
            g.lp[b] = lp;
        }
    }
    FPROF(0, 8);
}

template <int C0, int C1>
__global__ __launch_bounds__(FT) void actor_fwd_kernel(ActorFwdArgs g)
{
    __shared__ TileLds lds;
    actor_fwd_body<C0, C1>(g, lds, (int)blockIdx.x, (int)blockIdx.y);
}

// Launch (1) -- the actor on the next state, its second layer split over `ay` workgroups per tile -- and the policy-gradient sample's forward
// (unsplit: its activations are kept for the backward pass) as ONE launch: rows y < ay of the grid run the first, row ay the second.  The
// two share nothing (the sample's forward reads its state rows from the ring itself), so this replaces the side stream that carried the
// second one next to launches (1)-(2): one queue instead of two for most of the step (round 6; ERL_SAC_PAIR=0 keeps the side stream).
template <int C0, int C1A, int C1B>
__global__ __launch_bounds__(FT) void actor_fwd_pair_kernel(ActorFwdArgs ga, ActorFwdArgs gb, int ay)
{
    __shared__ TileLds lds;
    if ((int)blockIdx.y < ay) actor_fwd_body<C0, C1A>(ga, lds, (int)blockIdx.x, (int)blockIdx.y);
    else actor_fwd_body<C0, C1B>(gb, lds, (int)blockIdx.x, (int)blockIdx.y - ay);
}

// ---------------------------------------------------------------------------------------------------------
// critic passes: grid = (tiles, E).  MODE 0: forward only -> q[e][b] (the TARGET ensemble on the next state);
// MODE 1: training pass (labels, loss gradient, backward to the encoder output, operands of the weight gradients);
// MODE 2: policy-gradient pass (TARGET ensemble on [state | action_pg]: q and d(mean q)/d(action)).
// ---------------------------------------------------------------------------------------------------------
struct CriticArgs {
    const float *P;                          // critic or target parameter block
    FusedDims d;
    const float *Xs, *Xa;                    // (B, S), (B, A)
    float *q;                                // [E][B]: this pass's q values
    // MODE 1
    const float *qt;                         // [E][B] target q of the next state
    const float *reward, *undone, *unmask, *lp_next, *is_weight, *alpha0;
    float gamma;
    float *label, *dq;                       // (B,), [E][B]
    float *xa, *enc, *H1e, *dZ1e, *dEncE;    // (B, S+A), (B, h0), [E](B, h1), [E](B, h1), [E](B, h0)
    // MODE 2
    float *dAct, *qpart;                     // [E * split](B, A), [E * split][tiles]
    const float *qc, *label_in;              // the training pass's q and labels: the critic objective is finished here
    float *td_out, *tdpart;                  // (B,) or NULL, [tiles]
    // Feature split of the decoders (MODE 0 and 2: passes whose backward does not need q): grid.y = E * split, workgroup (e, s) owns
    // hidden features [s d.h1, (s + 1) d.h1) of decoder e -- d.h1 is the SLICE's width here -- i.e. rows of W1, entries of b1 and of the
    // output row, and streams a split-th of the decoder's weights.  What leaves is linear in the slice: q (bias from slice 0), the
    // logged sums and d(mean q)/d(action) come out as `split` partial results per decoder, added in slice order by their consumers.
    int split;
    int qt_split;                            // MODE 1: qt holds [E][qt_split][B] partial target values
    unsigned long long *span;                // measurement hook (erl_common.h: erl_span_*; set for the training pass); nullptr = off
    // MODE 1 with split > 1 (round 6).  The training pass's backward needs the FULL q of its own forward, so the `split` workgroups of a
    // (tile, decoder) exchange their shares of q inside the launch: each publishes {share, nonce} as ONE 8-byte agent-scope granule per
    // sample in qx[E * split][B] (library-owned, never cleared: the per-launch nonce invalidates older contents) and reads the others'
    // -- a bounded wait among workgroups that are all resident (tiles * E * split <= 256, three fit a CU) -- then every one of them adds
    // the shares in slice order (the same bits everywhere) and carries on with its own 64 hidden features: dZ1 and its slice of the
    // decoder's weight-gradient operands go straight to their places in the full-width matrices (h1_full), the share of dEnc = W1^T dZ1
    // goes to dEncP[E * split](B, h0) and the LAST of the four to arrive (arrive[tile * E + e], re-armed by it) adds the shares in slice
    // order into dEncE.  A workgroup streams 64 KB of the decoder (forward) + 64 KB (transposed) instead of 256 + 256 KB through one CU.
    int h1_full;                             // the decoder's hidden width when d.h1 is a slice's (0: d.h1)
    unsigned long long *qx;
    uint32_t nonce, spin_limit;
    float *dEncP;
    unsigned *arrive;
    uint32_t *fault;
};

// (ERL_SAC_TRAIN_WPE=4 compiles the training pass on decoder SLICES -- MODE 1, C1 = 0 -- for two workgroups per CU: 128 registers, 52 bytes
// of scratch.  That was needed while the policy-gradient sample's kernels ran next to it -- its 256 workgroups wait for each other's
// shares of q and the late ones kept their partners spinning: 33-35 us a launch; since that sample is forked before launch (1) the pass
// has the chip to itself and the uncapped build -- 149 registers, one workgroup per CU, no scratch -- is the faster one: 20.7 against
// 22.0 us, 139.7 against 142.5 us per update, profiles/r06_sac_train_split_ab.txt)
template <int MODE, int C0, int C1>
#ifndef ERL_SAC_TRAIN_WPE
#define ERL_SAC_TRAIN_WPE 2
#endif
__global__ __launch_bounds__(FT) __attribute__((amdgpu_waves_per_eu((MODE == 1 && C1 == 0) ? ERL_SAC_TRAIN_WPE : 2))) void critic_tile_kernel(CriticArgs g)
{
    __shared__ TileLds lds;
    __shared__ float dql[TS];
    const unsigned long long t_span = MODE == 1 ? erl_span_in(g.span) : 0ull;
    const LaneId L = lane_id();
    const FusedDims &d = g.d;
    // MODE 1 with slices: the `split` workgroups of a (tile, decoder) are CONSECUTIVE in dispatch order (they wait for each other: they
    // should become resident together), i.e. the slice is the fastest index of the linear workgroup id
    int bx = blockIdx.x, ey = blockIdx.y;
    if (MODE == 1 && g.split > 1) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.y, sl = lin % g.split, rest = lin / g.split;
        bx = rest % gridDim.x;
        ey = (rest / gridDim.x) * g.split + sl;
    }
    const int e = ey / g.split, E = d.E;
    const int64_t fo = (int64_t)(ey - e * g.split) * d.h1;          // first hidden feature of this workgroup's slice
    const int64_t B = d.B, row0 = (int64_t)bx * TS, row = row0 + L.l15;
    const bool valid = row < B;
    const float *Pd = g.P + d.cdec0 + (int64_t)e * d.dec;
    const bool first = ey == 0;
    FwdW<4, WClass<C0>::NU> we;
    FwdW<WClass<C0>::KT, WClass<C1>::NU> w1;
    SmallW wo;
    layer_fwd_load<4, WClass<C0>::NU, false>(g.P + d.cWe, d.S + d.A, d.h0, L, we);
    layer_fwd_load<WClass<C0>::KT, WClass<C1>::NU, true>(Pd + d.dW1 + fo * d.h0, d.h0, d.h1, L, w1);
    layer_small_load<false, true>(Pd + d.dWo + fo, d.h1, 1, 0, 0, L, wo);
    // (round 6, last session) what the epilogue reads of EARLIER launches' results -- the target's partial q values, the label's inputs, the
    // training pass's q for the td errors -- is requested here, behind the weights, instead of where it is used: there each was a round
    // trip of its own on the tile's critical path ("labels, dq": 10 % of the training pass in tools/sac_fused_profile.py's stamps).  Clamped
    // addresses, values selected where they are used; the same arithmetic in the same order.
    float pq[kQxSplit] = {0.f, 0.f, 0.f, 0.f}, p_rew = 0.f, p_und = 0.f, p_lpn = 0.f, p_unm = 0.f, p_w = 1.f, p_al0 = 0.f;
    float p_qe[FMAXE] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, p_lab = 0.f;
    if (MODE == 1) {
        if (L.tid < TS * FMAXE) {
            const int s_ = L.tid / FMAXE, k = min(L.tid - s_ * FMAXE, E - 1);
            const int64_t b = min(row0 + s_, B - 1);
#pragma unroll
            for (int j = 0; j < kQxSplit; ++j) pq[j] = g.qt[((size_t)k * g.qt_split + min(j, g.qt_split - 1)) * B + b];
        }
        if (L.tid < TS) {
            const int64_t b = min(row0 + L.tid, B - 1);
            p_rew = g.reward[b]; p_und = g.undone[b]; p_lpn = g.lp_next[b]; p_unm = g.unmask[b]; p_al0 = g.alpha0[0];
            if (g.is_weight) p_w = g.is_weight[b];
        }
    }
    if (MODE == 2 && first && L.wave == 1 && L.lane < TS) {
        const int64_t b = min(row0 + L.lane, B - 1);
        p_lab = g.label_in[b]; p_unm = g.unmask[b];
        if (g.is_weight) p_w = g.is_weight[b];
#pragma unroll
        for (int k = 0; k < FMAXE; ++k) p_qe[k] = g.qc[(size_t)min(k, E - 1) * B + b];
    }
    __builtin_amdgcn_sched_barrier(0);
    clear_images(lds.T0, lds.T1, L);
    lds_barrier();
    if (MODE == 1) FPROF(1, 0);
    load_rows(g.Xs, d.S, g.Xa, d.A, row0, B, lds.T0, (MODE == 1 && first) ? g.xa : nullptr, L);
    lds_barrier();
    if (MODE == 1) FPROF(1, 1);
    f32x4 z[2], gk[2], gk1[2];
    layer_fwd_mma<4, WClass<C0>::NU, false>(we, g.P + d.cbe, d.S + d.A, d.h0, lds.T0, L, z);            // shared encoder: raw linear
    emit_hidden(z, d.h0, false, lds.T1, gk, (MODE == 1 && first) ? g.enc : nullptr, nullptr, row, valid, L);
    lds_barrier();
    if (MODE == 1) FPROF(1, 2);
    layer_fwd_mma<WClass<C0>::KT, WClass<C1>::NU, true>(w1, Pd + d.db1 + fo, d.h0, d.h1, lds.T1, L, z);
    // the backward pass's weights are requested now (the forward layer's registers are free): they land under the output layer,
    // the loss and the gate
    BwdW<WClass<C1>::KT, WClass<C0>::NU> wb;
    SmallW wa;
    const int h1f = g.h1_full ? g.h1_full : d.h1;
    emit_hidden(z, d.h1, true, lds.T0, gk1, MODE == 1 ? g.H1e + (size_t)e * B * h1f + fo : nullptr, nullptr, row, valid, L, h1f);
    if (MODE != 0) layer_bwd_load<WClass<C1>::KT, WClass<C0>::NU>(Pd + d.dW1 + fo * d.h0, d.h1, d.h0, L, wb);
    if (MODE == 2) layer_small_load<true, false>(g.P + d.cWe, d.h0, d.A, d.S + d.A, d.S, L, wa);
    lds_barrier();
    if (MODE == 1) FPROF(1, 3);
    layer_small_mma<false, true>(wo, fo == 0 ? Pd + d.dbo : nullptr, d.h1, 1, lds.T0, lds.part, lds.Yl, L);
    if (MODE == 1) FPROF(1, 4);
    if (MODE == 0) {
        if (L.tid < TS && row0 + L.tid < B) g.q[(size_t)ey * B + row0 + L.tid] = lds.Yl[L.tid * 16];
        return;
    }
    if (MODE == 1 && g.split > 1) {
        // ---- the slices' shares of q meet here (CriticArgs::qx): publish, then read all `split` of them in slice order
        if (L.tid < TS && row0 + L.tid < B) {
            const int64_t b = row0 + L.tid;
            __hip_atomic_store(g.qx + (size_t)ey * B + b, (unsigned long long)__float_as_uint(lds.Yl[L.tid * 16]) | ((unsigned long long)g.nonce << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float qv = 0.f;
            // (the first look at every share goes out together: one round trip when the slices finish together, not one per slice; a share
            // that is not there yet is polled on its own, and the shares are added in slice order either way: the same bits)
            unsigned long long gr4[kQxSplit];
#pragma unroll
            for (int j = 0; j < kQxSplit; ++j)
                gr4[j] = __hip_atomic_load(g.qx + (size_t)(e * g.split + min(j, g.split - 1)) * B + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < kQxSplit; ++j) {
                if (j >= g.split) continue;
                const unsigned long long *src = g.qx + (size_t)(e * g.split + j) * B + b;
                unsigned long long gr = gr4[j];
                for (uint32_t spins = 0; (uint32_t)(gr >> 32) != g.nonce && spins < g.spin_limit; ++spins) {
                    __builtin_amdgcn_s_sleep(1);
                    gr = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const bool ready = (uint32_t)(gr >> 32) == g.nonce;
                // bounded: a share that never arrives poisons this sample's q with NaN AND is counted in the host-visible fault word
                if (!ready && g.fault) __hip_atomic_fetch_add(g.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                qv += ready ? __uint_as_float((uint32_t)gr) : __uint_as_float(0x7FC00000u);
            }
            lds.Yl[L.tid * 16] = qv;
        }
    }
    if (MODE == 1) FPROF(1, 5);
    // ---- the target's q of the next state, E values per sample: thread (sample, k) adds the qt_split partial values the target pass left
    __shared__ float qtl[TS * FMAXE];
    if (MODE == 1) {
        if (L.tid < TS * FMAXE) {
            const int s_ = L.tid / FMAXE, k = L.tid - s_ * FMAXE;
            const int64_t b = row0 + s_;
            float qk = 0.f;
            if (k < E && b < B) {                       // (requested at the kernel's start, added in slice order)
                qk = pq[0];
#pragma unroll
                for (int j = 1; j < kQxSplit; ++j)
                    if (j < g.qt_split) qk += pq[j];
            }
            qtl[L.tid] = qk;
        }
        lds_barrier();
    }
    // ---- dL/dq of this decoder for the tile's samples
    if (L.tid < TS) {
        const int64_t b = row0 + L.tid;
        float dqv = 0.f;
        if (b < B) {
            const float qv = lds.Yl[L.tid * 16];
            if (MODE != 1 || g.split <= 1) g.q[(size_t)ey * B + b] = qv;
            else if (fo == 0) g.q[(size_t)e * B + b] = qv;             // (the full q: every slice holds the same bits, slice 0 stores them)
            if (MODE == 1) {
                // q_label = reward + undone * gamma * (min_e q_target - next_logprob * alpha)      (AgentSAC.py:52-55)
                float m = qtl[L.tid * FMAXE];
                for (int k = 1; k < E; ++k) m = fminf(m, qtl[L.tid * FMAXE + k]);
                const float alpha = expf(p_al0);
                const float lab = p_rew + (p_und * g.gamma) * (m - p_lpn * alpha);
                if (first) g.label[b] = lab;
                // td = mean_e (q - label)^2 * unmask; obj = mean_b (td w): dq = 2 (q - label) unmask w / (E B)   (:57-62)
                const float w = g.is_weight ? p_w : 1.f;
                dqv = 2.f * (qv - lab) * p_unm * w / ((float)E * (float)B);
                if (fo == 0) g.dq[(size_t)e * B + b] = dqv;
            } else {
                dqv = -1.0f / ((float)E * (float)B);               // L = -(mean_b mean_e q - alpha mean_b logprob)   (:82-84)
            }
        }
        dql[L.tid] = dqv;
    }
    if (MODE == 2) {                                               // partial sums of q for the logged actor objective
        float s = (L.tid < TS && row0 + L.tid < B) ? lds.Yl[L.tid * 16] : 0.f;
        s = wave_sum(s);
        if (L.tid == 0) g.qpart[(size_t)ey * gridDim.x + blockIdx.x] = s;
        if (first && L.wave == 1) {                                // (wave 1, whole wave) the critic objective's per-sample td errors, now
            const int64_t b = row0 + L.lane;                       // that every decoder's q of the training pass is in memory
            float td = 0.f;
            if (L.lane < TS && b < B) {
                const float lab = p_lab;                    // (every decoder's q and the label requested at the kernel's start; the squares added in decoder order)
                float s2 = 0.f;
#pragma unroll
                for (int k = 0; k < FMAXE; ++k) {
                    if (k >= E) continue;
                    const float diff = p_qe[k] - lab;
                    s2 += diff * diff;
                }
                td = (s2 / (float)E) * p_unm;
                if (g.td_out) g.td_out[b] = td;
                td *= g.is_weight ? p_w : 1.f;
            }
            td = wave_sum(td);
            if (L.lane == 0) g.tdpart[blockIdx.x] = td;
        }
    }
    lds_barrier();
    if (MODE == 1) FPROF(1, 6);
    // ---- dZ1 = (Wo^T dq) * GELU'(z1): the wave's own tiles (it still holds their GELU')
    {
        const int NT = (d.h1 + 15) >> 4;
        const float dqs = dql[L.l15];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ot = L.wave + FWV * u;
            if (ot >= NT) continue;
            const int f = 16 * ot + 4 * L.q;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (f + r < d.h1) ? Pd[d.dWo + fo + f + r] * dqs * gk1[u][r] : 0.f;
            *reinterpret_cast<float4 *>(lds.T1 + L.l15 * LDT + f) = make_float4(v[0], v[1], v[2], v[3]);
            if (MODE == 1 && valid && f < d.h1)
                *reinterpret_cast<float4 *>(g.dZ1e + ((size_t)e * B + row) * h1f + fo + f) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    lds_barrier();
    if (MODE == 1) FPROF(1, 7);
    // ---- dEnc = W1^T dZ1 (the encoder is a raw linear layer: no gate)
    f32x4 dx[2];
    layer_bwd_mma<WClass<C1>::KT, WClass<C0>::NU>(wb, d.h1, d.h0, lds.T1, L, dx);
    {
        const int NT = (d.h0 + 15) >> 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = L.wave + FWV * u;
            if (it >= NT) continue;
            const int f = 16 * it + 4 * L.q;
            if (MODE == 1) {
                if (valid && f < d.h0) {
                    if (g.split > 1) {                           // this slice's share: a 16-byte write-through (`sc1`) store, past the per-XCD L2
                        st4_sc1(g.dEncP + ((size_t)ey * B + row) * d.h0 + f, f32x4{dx[u][0], dx[u][1], dx[u][2], dx[u][3]});
                    } else {
                        *reinterpret_cast<float4 *>(g.dEncE + ((size_t)e * B + row) * d.h0 + f) = make_float4(dx[u][0], dx[u][1], dx[u][2], dx[u][3]);
                    }
                }
            } else {
                *reinterpret_cast<float4 *>(lds.T0 + L.l15 * LDT + f) = make_float4(dx[u][0], dx[u][1], dx[u][2], dx[u][3]);
            }
        }
    }
    if (MODE == 1) FPROF(1, 8);
    if (MODE == 1) {
        if (g.split > 1) {
            // the last of the (tile, decoder)'s workgroups to arrive adds the shares of dEnc in slice order (actor_fwd_kernel's idiom)
            __shared__ int s_last1;
            // (16-byte `sc1` stores and loads spelled out: float-by-float agent-scope atomics cost 32 dependent round trips in the last
            // workgroup -- 45 us a launch --, release / acquire FENCES a whole-L2 write-back and invalidate per wave -- 70 us)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (L.tid == 0) {
                unsigned *cnt = g.arrive + (size_t)bx * E + e;
                const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last1 = old == (unsigned)g.split - 1u;
                if (s_last1) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-armed for the next launch
            }
            __syncthreads();
            if (s_last1) {
                const int NT = (d.h0 + 15) >> 4;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int it = L.wave + FWV * u;
                    const int f = 16 * it + 4 * L.q;
                    if (it >= NT || !valid || f >= d.h0) continue;
                    static_assert(kQxSplit == 4, "the asm below fetches four shares");
                    const float *p0 = g.dEncP + ((size_t)(e * kQxSplit) * B + row) * d.h0 + f;
                    const size_t sl = (size_t)B * d.h0;
                    f32x4 s0, s1, s2, s3;
                    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                                 "global_load_dwordx4 %1, %5, off sc1\n\t"
                                 "global_load_dwordx4 %2, %6, off sc1\n\t"
                                 "global_load_dwordx4 %3, %7, off sc1\n\t"
                                 "s_waitcnt vmcnt(0)"
                                 : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
                                 : "v"(p0), "v"(p0 + sl), "v"(p0 + 2 * sl), "v"(p0 + 3 * sl)
                                 : "memory");
                    const f32x4 acc = ((s0 + s1) + s2) + s3;                   // slice order
                    *reinterpret_cast<float4 *>(g.dEncE + ((size_t)e * B + row) * d.h0 + f) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                }
            }
        }
        FPROF(1, 9);
        erl_span_out(g.span, t_span);
        return;
    }
    lds_barrier();
    // ---- d(mean q)/d(action) = the action columns of We^T dEnc
    layer_small_mma<true, false>(wa, nullptr, d.h0, d.A, lds.T0, lds.part, lds.Yl, L);
    if (L.tid < TS * 16) {
        const int s = L.tid >> 4, a = L.tid & 15;
        if (a < d.A && row0 + s < B) g.dAct[((size_t)ey * B + row0 + s) * d.A + a] = lds.Yl[s * 16 + a];
    }
}

// ---------------------------------------------------------------------------------------------------------
// actor backward: head (tanh, reparameterisation, the log-prob-at-the-mean quirk) -> dZ2 -> dZ1; also finishes the two logged
// objectives from the partial sums of the passes before it (workgroup 0, fixed order)
// ---------------------------------------------------------------------------------------------------------
struct ActorBwdArgs {
    const float *P;
    FusedDims d;
    const float *Y, *act_t, *eps, *dAct, *alpha_log, *G0, *G1, *lp_cur;
    float *dY, *dZ2, *dZ1;                   // (B, 2A), (B, h1), (B, h0)
    const float *tdpart, *qpart;
    int ntiles;
    int nsplit;                              // dAct / qpart hold E * nsplit partial results (critic_tile_kernel's feature split)
    // Feature split of the last step, dH0 = W2^T dZ2 (output-split: no exchange): grid.y = split, workgroup (tile, s) owns first-layer
    // features [s d.h0, (s + 1) d.h0) -- d.h0 is the slice's width, h0_full the layer's -- i.e. 64 of W2's 256 columns; everything before it
    // (head backward, dZ2) is small and done by every slice, stored by slice 0.
    int split, h0_full;
    float *objs_out;
};

template <int C0, int C1>
__global__ __launch_bounds__(FT) void actor_bwd_kernel(ActorBwdArgs g)
{
    __shared__ TileLds lds;
    const LaneId L = lane_id();
    const FusedDims &d = g.d;
    const int A = d.A, E = d.E;
    const int64_t B = d.B, row0 = (int64_t)blockIdx.x * TS, row = row0 + L.l15;
    const bool valid = row < B;
    const int64_t fo = (int64_t)blockIdx.y * d.h0;                 // first first-layer feature of this workgroup's slice (0 unsplit)
    const int h0f = g.split > 1 ? g.h0_full : d.h0;
    const bool s0 = blockIdx.y == 0;
    if (blockIdx.x == 0 && s0) {                                   // the logged objectives (AgentSAC.py:86)
        float sl = 0.f, sqp = 0.f;
        for (int64_t i = L.tid; i < B; i += FT) sl += g.lp_cur[i];
        for (int k = L.tid; k < E * g.nsplit * g.ntiles; k += FT) sqp += g.qpart[k];      // (one thread walking the table: a load latency per entry)
        const float tl = block_sum(sl, lds.red);
        const float sq = block_sum(sqp, lds.red);
        if (L.tid == 0) {
            float std_ = 0.f;
            for (int t0 = 0; t0 < g.ntiles; t0 += 16) {      // (sixteen entries per round trip instead of one; added in tile order)
                float tv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) tv[u] = g.tdpart[min(t0 + u, g.ntiles - 1)];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (t0 + u < g.ntiles) std_ += tv[u];
            }
            g.objs_out[0] = std_ / (float)B;
            g.objs_out[1] = sq / ((float)E * (float)B) - expf(g.alpha_log[0]) * (tl / (float)B);
        }
    }
    BwdW<1, WClass<C1>::NU> wbh;
    BwdW<WClass<C1>::KT, WClass<C0>::NU> wb2;
    layer_bwd_load<1, WClass<C1>::NU>(g.P + d.aWh, 2 * A, d.h1, L, wbh);
    clear_images(lds.T0, lds.T1, L);
    // d(mean q)/d(action) arrives as E * nsplit partial results per (sample, action): thread (sample, action, j) adds every fourth of them,
    // the head thread below adds the four sums in order (E = 4 unsplit: the former left-to-right sum)
    __shared__ float dal[TS * 8 * 4];
    {
        const int s_ = L.tid >> 5, a = (L.tid >> 2) & 7, j = L.tid & 3, nk = E * g.nsplit;
        const int64_t b = row0 + s_;
        float acc = 0.f;
        if (a < A && b < B)
            for (int k = j; k < nk; k += 4) acc += g.dAct[((size_t)k * B + b) * A + a];
        dal[L.tid] = acc;
    }
    lds_barrier();
    if (L.tid < TS) {                                              // dL/d(head output) of sample row0 + tid (head_backward_kernel, sac.hip)
        const int64_t b = row0 + L.tid;
        if (b < B) {
            const float alpha = expf(g.alpha_log[0]);
            const float dlp = alpha / (float)B;
            for (int a = 0; a < A; ++a) {
                const float ls = g.Y[b * 2 * A + A + a];
                const float lsc = fminf(fmaxf(ls, -16.f), 2.f);
                const float sd = expf(lsc);
                const float t = g.act_t[b * A + a];
                const float one_m = 1.f - t * t;
                const float *dp = dal + (L.tid * 8 + a) * 4;
                const float dA = ((dp[0] + dp[1]) + dp[2]) + dp[3];
                const float du = dA * one_m + dlp * (2.f * t * one_m / (one_m + 1e-6f));
                const bool inside = ls >= -16.f && ls <= 2.f;
                const float dls = inside ? du * sd * g.eps[b * A + a] - dlp : 0.f;
                lds.T0[L.tid * LDT + a] = du;
                lds.T0[L.tid * LDT + A + a] = dls;
                if (s0) {
                    g.dY[b * 2 * A + a] = du;
                    g.dY[b * 2 * A + A + a] = dls;
                }
            }
        }
    }
    lds_barrier();
    layer_bwd_load<WClass<C1>::KT, WClass<C0>::NU>(g.P + d.aW2 + fo, d.h1, d.h0, L, wb2, h0f);   // (lands under the head layer and its gate)
    f32x4 dx[2];
    layer_bwd_mma<1, WClass<C1>::NU>(wbh, 2 * A, d.h1, lds.T0, L, dx);        // dH1 = Wh^T dY
    {
        const int NT = (d.h1 + 15) >> 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = L.wave + FWV * u;
            if (it >= NT) continue;
            const int f = 16 * it + 4 * L.q;
            float4 gate = zero4();
            if (valid && f < d.h1) gate = *reinterpret_cast<const float4 *>(g.G1 + row * d.h1 + f);
            const float4 v = make_float4(dx[u][0] * gate.x, dx[u][1] * gate.y, dx[u][2] * gate.z, dx[u][3] * gate.w);
            *reinterpret_cast<float4 *>(lds.T1 + L.l15 * LDT + f) = v;
            if (s0 && valid && f < d.h1) *reinterpret_cast<float4 *>(g.dZ2 + row * d.h1 + f) = v;
        }
    }
    lds_barrier();
    layer_bwd_mma<WClass<C1>::KT, WClass<C0>::NU>(wb2, d.h1, d.h0, lds.T1, L, dx);   // dH0 = W2^T dZ2
    {
        const int NT = (d.h0 + 15) >> 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = L.wave + FWV * u;
            if (it >= NT) continue;
            const int f = 16 * it + 4 * L.q;
            if (valid && f < d.h0) {
                const float4 gate = *reinterpret_cast<const float4 *>(g.G0 + row * h0f + fo + f);
                *reinterpret_cast<float4 *>(g.dZ1 + row * h0f + fo + f) =
                    make_float4(dx[u][0] * gate.x, dx[u][1] * gate.y, dx[u][2] * gate.z, dx[u][3] * gate.w);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// all weight / bias gradients of a network in one launch: a table of contractions over the batch
//     dW[m][n] = sum_b (sum_k dZ_k[b][m]) X[b][n],   db[m] = sum_b (sum_k dZ_k[b][m])
// One workgroup (4 waves) per 32 x 32 tile of one dW: the waves split the batch rows, each contracts its quarter on
// v_mfma_f32_32x32x2_f32 with both operands straight from memory (16 row pairs in flight), the four partial tiles meet in LDS and
// are added in wave order.  The bias gradient rides the A operand of the tile column 0.
// ---------------------------------------------------------------------------------------------------------
constexpr int DW_MAXP = 1 + 2 * FMAXE;      // the critic's table: the shared encoder + two layers per decoder (round 6: was 12 -- eight critics
                                           // wrote five problems past the table on the host: found by tests/test_sac.py's E = 8 case)
struct DwProb {
    const float *dZ; int64_t sZ; int nZ; int M;     // (B, M) row-major, nZ matrices sZ floats apart summed in order
    const float *X; int N;                          // (B, N) row-major
    float *dW, *db;                                 // [M][N], [M]
    int tiles_n, tile0, ntiles;
};
struct DwArgs {
    DwProb p[DW_MAXP];
    int np;
    int64_t B;
    float *clamp_alpha_log;     // not NULL: workgroup 0 also clamps the temperature's logarithm to [-16, 2] (see erl_sac_update_fused)
    double *norm_parts;         // not NULL: [workgroups] the fp64 sum of squares of what each workgroup stores (its dW tile, its db rows): the
                                // squared gradient norm in pieces, so that clip + Adam needs no pass over the gradient and no grid-wide wait
    // not NULL (al_lp): workgroup 0 first takes the temperature's Adam step (alpha_step_block: obj_alpha, AgentSAC.py:76-79) -- the critic's
    // table carries it in the one-stream form of the step (round 6): nothing between launch (1) and the actor's backward reads alpha_log
    const float *al_lp;
    int64_t al_n;
    float *al_alpha_log, *al_m1, *al_m2;
    float al_target_entropy, al_beta1, al_beta2, al_eps, al_max_norm, al_step_size, al_bc2_sqrt;
    int xcd_tiles;              // not 0: the launch's tile count, and workgroups take their tiles in the XCD-contiguous order (dw_table_kernel)
    int alpha_wg;               // the workgroup that carries the temperature's duties: 0 (with its tile), or the extra one behind the last tile
};
constexpr int kDwMaxParts = 1024;      // workgroups of one dw_table launch at the widest supported network (8 decoders of 256 x 256: 592)

// obj_alpha = mean(alpha_log * (target_entropy - logprob)): g = target_entropy - mean(logprob); clip + Adam on one element (the arithmetic of
// alpha_step_kernel, sac.hip).  One 256-thread workgroup; every thread of it must call.
__device__ __forceinline__ void alpha_step_block(const float *__restrict__ lp, int64_t n, float target_entropy, float *__restrict__ alpha_log,
                                                 float *__restrict__ m1, float *__restrict__ m2, float beta1, float beta2, float eps, float max_norm,
                                                 float step_size, float bc2_sqrt)
{
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += lp[i];
    const float t = block_sum(s, red);
    if (threadIdx.x != 0) return;
    const float gr = t * (-1.0f / (float)n) + target_entropy;
    const float total_norm = (float)sqrt((double)gr * (double)gr);
    float coef = max_norm / (total_norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    float e_m1 = m1[0], e_m2 = m2[0], e_p = alpha_log[0];
    erl_adam_update(erl_mul_rn(gr, erl_mul_rn(1.0f, coef)), e_m1, e_m2, e_p, beta1, beta2, eps, step_size, bc2_sqrt);   // the library's one Adam
    m1[0] = e_m1;
    m2[0] = e_m2;
    alpha_log[0] = e_p;
}

template <int U, int KC>
__global__ __launch_bounds__(256) void dw_table_kernel(DwArgs g)
{
    __shared__ float red[4][32 * 33];
    __shared__ float bsum[4][32];
    __shared__ double nscratch[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    // the temperature's duties: workgroup 0's first duty (default), or a workgroup of their own behind the last tile (ERL_SAC_DW=2: the serial
    // prefix -- a sum over the batch, then a dependent Adam round trip -- leaves workgroup 0; measured even, the launch is not held by it)
    const int wg_alpha = g.alpha_wg;                              // 0, or the extra workgroup's index
    // which tile: blockIdx.x itself, or (g.xcd_tiles: the launch's tile count) the XCD-contiguous order -- consecutive workgroups go to
    // consecutive XCDs, so workgroup b takes tile (b % 8) * (tiles / 8) + b / 8 (remainders spread over the first XCDs): the tiles one
    // XCD's L2 serves are then neighbours in (problem, tm, tn) order and share their dZ row blocks and X column blocks, instead of every XCD
    // fetching every block of every problem over the fabric.  Everything below (the squared-norm piece too) is indexed by the TILE: same bits.
    int bid = blockIdx.x;
    if (g.xcd_tiles && bid < g.xcd_tiles) {
        const int x = bid & 7, i = bid >> 3, fl = g.xcd_tiles >> 3, rem = g.xcd_tiles & 7;
        bid = x * fl + min(x, rem) + i;
    }
    if (g.clamp_alpha_log && bid == wg_alpha && tid == 0)        // after alpha was read (AgentSAC.py:80-81): the actor's backward has run
        g.clamp_alpha_log[0] = fminf(fmaxf(g.clamp_alpha_log[0], -16.f), 2.f);
    if (g.al_lp && bid == wg_alpha)                              // (workgroup-uniform: every thread of that workgroup takes part in the sum)
        alpha_step_block(g.al_lp, g.al_n, g.al_target_entropy, g.al_alpha_log, g.al_m1, g.al_m2, g.al_beta1, g.al_beta2, g.al_eps, g.al_max_norm,
                         g.al_step_size, g.al_bc2_sqrt);
    if (wg_alpha && bid == wg_alpha) return;                     // (the extra workgroup owns no tile and no squared-norm piece)
    int pi = 0;
    for (int k = 1; k < g.np; ++k)
        if (bid >= g.p[k].tile0) pi = k;
    const DwProb &p = g.p[pi];
    const int t = bid - p.tile0, tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int m = 32 * tm + l31, n = 32 * tn + l31;
    const bool mok = m < p.M, nok = n < p.N;
    const int64_t B = g.B;
    const int64_t per = ((B + 7) / 8) * 2;                          // rows per wave (even)
    const int64_t b0 = per * wave, b1 = min(B, b0 + per);
    f32x16 acc = {0};
    float bs = 0.f;
    const int mc = min(m, p.M - 1), nc = min(n, p.N - 1);
    const int64_t rmax = B - 1;
    // U row pairs per round trip; the nZ - 1 further matrices of a summed A operand (the encoder's gradient: one dEnc per decoder) are
    // requested KC at a time in the same round trip (round 6, last session: one more round trip per further matrix made the encoder's
    // tiles -- six round trips at four decoders against two -- the launch's stragglers).  Same loads, same order of every sum and of the
    // MFMAs in every instantiation: the same bits.
    for (int64_t b = b0; b < b1; b += 2 * U) {       // (uniform trip count; loads clamped + selected: 2 U .. (2 + KC) U in flight per lane)
        float a[U], x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rc = min(b + 2 * u + hi, rmax);
            a[u] = p.dZ[rc * p.M + mc];
            x[u] = p.X[rc * p.N + nc];
        }
        for (int k = 1; k < p.nZ; k += KC) {          // A = the sum of nZ matrices, added in order (the encoder gradient: sum over decoders)
            float t[KC][U];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int kk = min(k + c, p.nZ - 1);
#pragma unroll
                for (int u = 0; u < U; ++u) t[c][u] = p.dZ[(size_t)kk * p.sZ + min(b + 2 * u + hi, rmax) * p.M + mc];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (k + c < p.nZ) {
#pragma unroll
                    for (int u = 0; u < U; ++u) a[u] += t[c][u];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);           // (all loads issued; masks afterwards: see ldw4_raw)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool rok = b + 2 * u + hi < b1;
            a[u] *= (rok && mok) ? 1.f : 0.f;
            x[u] *= (rok && nok) ? 1.f : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = mfma32(a[u], x[u], acc);
            bs += a[u];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][crow(r, hi) * 33 + l31] = acc[r];
    bs += __shfl_xor(bs, 32, 64);
    if (hi == 0) bsum[wave][l31] = bs;
    lds_barrier();
    double sq = 0.0;
    for (int e = tid; e < 32 * 32; e += 256) {
        const int i = e >> 5, j = e & 31;
        const float s = ((red[0][i * 33 + j] + red[1][i * 33 + j]) + red[2][i * 33 + j]) + red[3][i * 33 + j];
        if (32 * tm + i < p.M && 32 * tn + j < p.N) {
            p.dW[(size_t)(32 * tm + i) * p.N + 32 * tn + j] = s;
            sq += (double)s * (double)s;
        }
    }
    if (tn == 0 && tid < 32 && 32 * tm + tid < p.M && p.db) {
        const float s = ((bsum[0][tid] + bsum[1][tid]) + bsum[2][tid]) + bsum[3][tid];
        p.db[32 * tm + tid] = s;
        sq += (double)s * (double)s;
    }
    if (g.norm_parts) {                 // (fixed association: thread-strided, wave butterfly, waves in order)
        sq = block_sum(sq, nscratch);
        if (tid == 0) g.norm_parts[bid] = sq;
    }
}

int dw_add(DwArgs &a, const float *dZ, int64_t sZ, int nZ, int M, const float *X, int N, float *dW, float *db)
{
    if (a.np >= DW_MAXP) { a.np = DW_MAXP + 1; return -1; }          // (dw_launch refuses a table that overflowed)
    DwProb &p = a.p[a.np];
    p.dZ = dZ; p.sZ = sZ; p.nZ = nZ; p.M = M; p.X = X; p.N = N; p.dW = dW; p.db = db;
    p.tiles_n = (N + 31) / 32;
    p.ntiles = ((M + 31) / 32) * p.tiles_n;
    p.tile0 = a.np ? a.p[a.np - 1].tile0 + a.p[a.np - 1].ntiles : 0;
    return ++a.np;
}

int dw_launch(const DwArgs &a, hipStream_t s)
{
    ERL_REQUIRE(a.np >= 1 && a.np <= DW_MAXP, "erl_sac_update_f32(fused): weight-gradient table of %d problems", a.np);
    const DwProb &last = a.p[a.np - 1];
    ERL_REQUIRE(!a.norm_parts || last.tile0 + last.ntiles <= kDwMaxParts, "erl_sac_update_f32(fused): %d weight-gradient tiles, table of %d",
                last.tile0 + last.ntiles, kDwMaxParts);
    // ERL_SAC_DW (read at every launch: the forms are compared inside one process by the tests; all leave the same bits):
    //   0 (default) a summed operand's matrices requested in ONE round trip, three at a time (dw_table_kernel<16, 3>): config 3 118.7 -> 117.1 us
    //     per update, the launch 9.45 -> 8.66 us on average (profiles/r06_sac_dw_ab.txt)
    //   1 round 6's earlier form: one round trip per further matrix (<16, 1>)
    //   2 as 0, and the temperature's duties in a workgroup of their own behind the last tile instead of in front of workgroup 0's tile: measured even
    //   3 as 1 with that workgroup
    //   5 as 0 with the tiles handed out in XCD-contiguous order; 6 as 1 with it
    //   4 as 0 with a wave's whole quarter of a 256-sample batch in one round trip (<32, 3>: 325 registers, one wave per SIMD): measured slower
    const char *form_env = getenv("ERL_SAC_DW");
    const int form = form_env ? atoi(form_env) : 0;
    DwArgs b = a;
    const int ntiles = last.tile0 + last.ntiles;
    const bool own_wg = (form == 2 || form == 3) && (b.al_lp || b.clamp_alpha_log);
    b.alpha_wg = own_wg ? ntiles : 0;
    b.xcd_tiles = (form == 5 || form == 6) ? ntiles : 0;
    const dim3 grid(ntiles + (own_wg ? 1 : 0));
    if (form == 1 || form == 3 || form == 6) hipLaunchKernelGGL((dw_table_kernel<16, 1>), grid, dim3(256), 0, s, b);
    else if (form == 4) hipLaunchKernelGGL((dw_table_kernel<32, 3>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((dw_table_kernel<16, 3>), grid, dim3(256), 0, s, b);
    return erl_hip_status(hipGetLastError(), "erl_sac_update_f32(fused: dw_table)");
}

__global__ __launch_bounds__(256) void alpha_step_fused_kernel(const float *__restrict__ lp, int64_t n, float target_entropy, float *__restrict__ alpha_log,
                                                               float *__restrict__ m1, float *__restrict__ m2, float beta1, float beta2, float eps,
                                                               float max_norm, float step_size, float bc2_sqrt)
{
    alpha_step_block(lp, n, target_entropy, alpha_log, m1, m2, beta1, beta2, eps, max_norm, step_size, bc2_sqrt);
}

int wclass(int width) { return width <= 64 ? 0 : (width <= 128 ? 1 : 2); }

// a library-owned second stream per device (non-blocking) + the events that fork it from / join it into the caller's stream: the
// policy-gradient sample (actor forward on `state`) and the temperature step depend on nothing the critic update produces, so they
// run NEXT TO it (ERL_SAC_STREAMS=1 keeps everything on the caller's stream)
struct SacSide {
    int device = -1;
    hipStream_t owner = nullptr;          // the caller's stream this side stream serves
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, mid = nullptr;
    unsigned *arrive = nullptr;           // [256] arrival counters of the split actor forward (zero between launches), owned by this slot
    unsigned *arrive1 = nullptr;          // [256] ... of the split critic training pass (its dEnc shares)
    unsigned long long *qx = nullptr;     // [FMAXE * kQxSplit * 4096] granules {share of q, nonce}: the split critic training pass's q exchange
    uint32_t nonce = 0;
    unsigned long long *yx = nullptr;     // [2 passes][kQxSplit][4096][16] granules {share of the actor's head output, nonce} (ActorFwdArgs::yx)
    uint32_t nonce_y = 0;
};
constexpr size_t kYxPass = (size_t)kQxSplit * 4096 * 16;      // granules of one pass
SacSide g_sac_side[16];

// one side stream + event pair per (device, caller stream): two agents on different streams of one device never record each
// other's events
SacSide *sac_side_stream(hipStream_t owner)
{
    static const bool off = [] { const char *e = getenv("ERL_SAC_STREAMS"); return e && atoi(e) == 1; }();
    int dev = -1;
    if (off || hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    for (auto &q : g_sac_side)
        if (q.stream && q.device == dev && q.owner == owner) return &q;
    for (auto &q : g_sac_side) {
        if (q.stream) continue;
        if (hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&q.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q.join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q.mid, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            q.stream = nullptr;
            return nullptr;
        }
        void *cnt = nullptr;
        if (hipMalloc(&cnt, 512 * sizeof(unsigned)) == hipSuccess && hipMemset(cnt, 0, 512 * sizeof(unsigned)) == hipSuccess) {
            q.arrive = (unsigned *)cnt;
            q.arrive1 = q.arrive + 256;
        } else {
            (void)hipGetLastError();          // (no counters: the actor's forward and the critic's training pass stay unsplit)
        }
        void *qx = nullptr;
        const size_t qx_bytes = (size_t)FMAXE * kQxSplit * 4096 * sizeof(unsigned long long);
        if (hipMalloc(&qx, qx_bytes) == hipSuccess && hipMemset(qx, 0, qx_bytes) == hipSuccess) q.qx = (unsigned long long *)qx;
        else (void)hipGetLastError();
        void *yx = nullptr;
        if (hipMalloc(&yx, 2 * kYxPass * sizeof(unsigned long long)) == hipSuccess && hipMemset(yx, 0, 2 * kYxPass * sizeof(unsigned long long)) == hipSuccess)
            q.yx = (unsigned long long *)yx;
        else (void)hipGetLastError();
        (void)hipDeviceSynchronize();             // (once per slot: the caller's stream may be non-blocking, i.e. not ordered behind those memsets)
        q.device = dev;
        q.owner = owner;
        return &q;
    }
    return nullptr;                          // every slot taken: run on the caller's stream
}

// after the fork every exit path must bring the side stream's work back under the caller's stream (an early error return would
// otherwise leave kernels running on the shared workspace while the caller moves on)
struct SacJoin {
    SacSide *side;
    hipStream_t s;
    bool armed = false;
    ~SacJoin()
    {
        if (armed && side) {
            (void)hipEventRecord(side->join, side->stream);
            (void)hipStreamWaitEvent(s, side->join, 0);
        }
    }
};

// ---------------------------------------------------------------------------------------------------------
// Persistent off-policy rollout for the device-resident SynVecEnv: ONE launch per AgentSAC.explore_env.
//
// Replaces the loop of AgentBase._explore_vec_env (elegantrl/agents/AgentBase.py:130-170): for t in range(H): ActorSAC.get_action
// (AgentSAC.py:179-185), `states[t] = state`, `actions[t] = action`, env.step, the reward / flag stores; then `rewards *= reward_scale`
// and the two logical_not.  Per step the loop cost an explore launch (actor_fwd_kernel: 21 us at 64 envs, almost all of it four
// workgroups streaming the actor's cold weights) and an env launch; here a workgroup owns a 16-env tile for all H steps and the
// actor's weights stay in REGISTERS (the A operands actor_fwd_kernel loads once per launch anyway: 32 + 128 + 8 per lane at
// [256,256]), Ws^T / Wa^T of the environment and the state tile in LDS.  A step is instruction for instruction actor_fwd_kernel
// followed by erl_synenv_step_f32's tile form (envs.hip; the code of rollout_fused.hip's SynVecEnv branch), so the five rollout
// tensors, the final state and the env's counters are bit-identical to the per-step loop under the same Philox keys / injected
// noise (tests/test_sac.py).
// ---------------------------------------------------------------------------------------------------------
struct SacRolloutArgs {
    const float *P;
    FusedDims d;                                   // B = number of envs
    int H;
    const float *noise;                            // (H, N, A) or NULL: Philox keyed by (seed, counter0 + t, env, a)
    uint64_t seed, counter0;
    float reward_scale;
    float *o_states, *o_actions, *o_rewards;       // (H, N, S), (H, N, A), (H, N)
    uint8_t *o_undones, *o_unmasks;                // (H, N): !terminal, !truncate
    float *o_last_state;                           // (N, S) or NULL
    float *env_state;                              // (N, S) live state (PendulumVecEnv: the observation cos, sin, theta_dot)
    const float *Ws, *Wa;                          // SynVecEnv
    float *phys;                                   // PendulumVecEnv: (N, 2) theta, theta_dot
    int32_t *step_count, *episode;
    int max_step;
    uint64_t env_seed;
};

constexpr int SR_XLD = 68, SR_WLD = 68;
// dynamic LDS (floats): [XS 16 x 68][WST 64 x 68][WAT 64 x 16][ACT 16 x 16][RED 8 x 16 x 2][W1L 256 x 68: the actor's first layer][BIA 3 x 256: b1 | b2 | head bias]
constexpr int SR_O_XS = 0, SR_O_WST = SR_O_XS + 16 * SR_XLD, SR_O_WAT = SR_O_WST + 64 * SR_WLD, SR_O_ACT = SR_O_WAT + 64 * 16;
constexpr int SR_O_RED = SR_O_ACT + 16 * 16, SR_O_W1L = SR_O_RED + 8 * 16 * 2, SR_O_BIA = SR_O_W1L + FMAXW * SR_WLD, SR_FLOATS = SR_O_BIA + 3 * FMAXW;

enum { SR_ENV_SYN = 0, SR_ENV_PENDULUM = 1 };

template <int C0, int C1, int ENV>
__global__ __launch_bounds__(FT) void sac_rollout_synenv_kernel(SacRolloutArgs g)
{
    __shared__ TileLds lds;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *XS = smem + SR_O_XS, *WST = smem + SR_O_WST, *WAT = smem + SR_O_WAT, *ACT = smem + SR_O_ACT, *RED = smem + SR_O_RED, *W1L = smem + SR_O_W1L, *BIA = smem + SR_O_BIA;
    const LaneId L = lane_id();
    const FusedDims &d = g.d;
    const int S = d.S, A = d.A, H = g.H, ns = (S + 15) >> 4;
    const int64_t N = d.B, row0 = (int64_t)blockIdx.x * TS, env = row0 + L.l15;
    const bool valid = env < N;
    const int64_t row = valid ? env : N - 1;              // rows past N replay env N - 1 (never stored)
    // the weights of the second layer and of the head, once; the first layer's (<= 64 columns) wait in LDS, zero padded -- 32 registers
    // less across the loop, which at [256,256] is what fits
    FwdW<WClass<C0>::KT, WClass<C1>::NU> w2;
    SmallW wh;
    layer_fwd_load<WClass<C0>::KT, WClass<C1>::NU, true>(g.P + d.aW2, d.h0, d.h1, L, w2);
    layer_small_load<false, true>(g.P + d.aWh, d.h1, 2 * d.A, 0, 0, L, wh);
    clear_images(lds.T0, lds.T1, L);
    for (int e = L.tid; e < 16 * 64; e += FT) {            // state tile (rows past N: env N - 1), zero beyond S
        const int i = e >> 6, k = e & 63;
        const int64_t r_ = min(row0 + i, N - 1);
        XS[i * SR_XLD + k] = (k < S) ? g.env_state[r_ * S + min(k, S - 1)] : 0.f;
    }
    if (ENV == SR_ENV_SYN) {
        for (int e = L.tid; e < 64 * 64; e += FT) {        // WST[j][k] = Ws[k][j]
            const int k = e >> 6, jj = e & 63;
            WST[jj * SR_WLD + k] = (k < S && jj < S) ? g.Ws[(size_t)min(k, S - 1) * S + min(jj, S - 1)] : 0.f;
        }
        for (int e = L.tid; e < 16 * 64; e += FT) {
            const int k = e >> 6, jj = e & 63;
            WAT[jj * 16 + k] = (k < A && jj < S) ? g.Wa[(size_t)min(k, A - 1) * S + min(jj, S - 1)] : 0.f;
        }
    }
    if (L.tid < 16 * 16) ACT[L.tid] = 0.f;
    for (int e = L.tid; e < FMAXW * 64; e += FT) {         // W1L[row][k] = W1[row][k], zero beyond (h0, S)
        const int i = e >> 6, k = e & 63;
        W1L[i * SR_WLD + k] = (i < d.h0 && k < S) ? g.P[d.aW1 + (size_t)min(i, d.h0 - 1) * S + min(k, S - 1)] : 0.f;
    }
    for (int e = L.tid; e < 3 * FMAXW; e += FT) {          // the biases too (read per step: hoisted out of the loop they would be 24 more registers)
        const int which = e / FMAXW, k = e - which * FMAXW;
        const int n = which == 0 ? d.h0 : (which == 1 ? d.h1 : 2 * A);
        BIA[e] = k < n ? g.P[(which == 0 ? d.ab1 : (which == 1 ? d.ab2 : d.abh)) + min(k, n - 1)] : 0.f;
    }
    const float *wst_row = WST + ((16 * L.wave + L.l15) & 63) * SR_WLD + 4 * L.q, *wat_row = WAT + ((16 * L.wave + L.l15) & 63) * 16 + 4 * L.q;
    int sc = g.step_count[row], ep = g.episode[row];
    // Pendulum: thread s < 16 steps env row0 + s by itself (no matrix product, no cross-wave reduction)
    const int64_t prow = min(row0 + (int64_t)min(L.tid, TS - 1), N - 1);
    float th = 0.f, thdot = 0.f;
    int psc = 0, pep = 0;
    if (ENV == SR_ENV_PENDULUM) { th = g.phys[2 * prow]; thdot = g.phys[2 * prow + 1]; psc = g.step_count[prow]; pep = g.episode[prow]; }
    lds_barrier();

    for (int t = 0; t < H; ++t) {
        FwdW<4, WClass<C0>::NU> w1;
#pragma unroll
        for (int u = 0; u < WClass<C0>::NU; ++u) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                w1.v[u][kt] = *reinterpret_cast<const float4 *>(W1L + (16 * (L.wave + FWV * u) + L.l15) * SR_WLD + 16 * kt + 4 * L.q);
        }
        // ---- the state rows into the first layer's input image; states[t] = state
        for (int e = L.tid; e < TS * 16 * ns; e += FT) {
            const int s_ = e / (16 * ns), c = e - s_ * (16 * ns);
            const float v = XS[s_ * SR_XLD + c];
            lds.T0[s_ * LDT + c] = v;
            if (c < S && row0 + s_ < N) g.o_states[((size_t)t * N + row0 + s_) * S + c] = v;
        }
        lds_barrier();
        f32x4 z[2], gk[2];
        layer_fwd_mma<4, WClass<C0>::NU, true>(w1, BIA, 16 * ns, d.h0, lds.T0, L, z);     // (the LDS image is zero padded: no masks; the same ns k-tiles)
        emit_hidden(z, d.h0, true, lds.T1, gk, nullptr, nullptr, row, valid, L);
        lds_barrier();
        layer_fwd_mma<WClass<C0>::KT, WClass<C1>::NU, true>(w2, BIA + FMAXW, d.h0, d.h1, lds.T1, L, z);
        emit_hidden(z, d.h1, true, lds.T0, gk, nullptr, nullptr, row, valid, L);
        lds_barrier();
        layer_small_mma<false, true>(wh, BIA + 2 * FMAXW, d.h1, 2 * d.A, lds.T0, lds.part, lds.Yl, L);
        if (L.tid < TS) {                                  // actor_fwd_kernel's tail: action = tanh(mean + std * eps)
            const int64_t b = row0 + L.tid, bc = min(b, N - 1);
            for (int a = 0; a < A; ++a) {
                const float mean = lds.Yl[L.tid * 16 + a], ls = lds.Yl[L.tid * 16 + A + a];
                const float lsc = fminf(fmaxf(ls, -16.f), 2.f);
                const float sd = expf(lsc);
                const float eps = g.noise ? g.noise[((size_t)t * N + bc) * A + a]
                                          : philox_normal(g.seed, g.counter0 + (uint64_t)t, (uint32_t)bc, (uint32_t)a);
                const float tv = tanhf(mean + sd * eps);
                ACT[L.tid * 16 + a] = tv;
                if (b < N) g.o_actions[((size_t)t * N + b) * A + a] = tv;
            }
            if (ENV == SR_ENV_PENDULUM) {
                // Pendulum-v1 behind the reference wrapper's scaling (envs.hip pendulum_step_kernel, operation for operation)
                const float PI = 3.14159265358979323846f;
                float u = 2.f * ACT[L.tid * 16];
                u = fminf(fmaxf(u, -2.f), 2.f);
                const float two_pi = 2.f * PI;
                float ang = fmodf(th + PI, two_pi);
                if (ang < 0.f) ang += two_pi;
                ang -= PI;
                const float cost = ang * ang + 0.1f * thdot * thdot + 0.001f * u * u;
                float nthdot = thdot + (3.f * 10.f / 2.f * sinf(th) + 3.f * u) * 0.05f;
                nthdot = fminf(fmaxf(nthdot, -8.f), 8.f);
                float nth = th + nthdot * 0.05f;
                const int sc1 = psc + 1;
                const bool trunc = sc1 >= g.max_step;
                if (trunc) {       // reset: theta ~ U(-pi, pi), theta_dot ~ U(-1, 1)
                    pep = pep + 1;
                    const Philox4 p = philox4x32_10((uint32_t)bc, 0u, (uint32_t)pep, 0x50454e44u, (uint32_t)g.env_seed, (uint32_t)(g.env_seed >> 32));
                    nth = ((float)(p.x >> 8) * (1.f / 16777216.f) * 2.f - 1.f) * PI;
                    nthdot = (float)(p.y >> 8) * (1.f / 16777216.f) * 2.f - 1.f;
                }
                th = nth;
                thdot = nthdot;
                psc = trunc ? 0 : sc1;
                XS[L.tid * SR_XLD + 0] = cosf(nth);
                XS[L.tid * SR_XLD + 1] = sinf(nth);
                XS[L.tid * SR_XLD + 2] = nthdot;
                if (b < N) {
                    const float rew = -0.5f * cost;
                    g.o_rewards[(size_t)t * N + b] = g.reward_scale == 1.0f ? rew : rew * g.reward_scale;
                    g.o_undones[(size_t)t * N + b] = 1;
                    g.o_unmasks[(size_t)t * N + b] = trunc ? 0 : 1;
                }
            }
        }
        lds_barrier();
        if (ENV == SR_ENV_PENDULUM) continue;              // (the new state tile is in place: the barrier above publishes it)
        // ---- env.step on the matrix cores: s' = s Ws + a Wa, wave w < ns owns features 16 w .. 16 w + 15 (envs.hip synenv_tile_kernel)
        float out[4] = {0.f, 0.f, 0.f, 0.f}, a2 = 0.f;
        const int j0 = 16 * L.wave + 4 * L.q;
        if (L.wave < ns) {
            const float4 av = *reinterpret_cast<const float4 *>(ACT + L.l15 * 16 + 4 * L.q);
            a2 = (av.x * av.x + av.y * av.y) + (av.z * av.z + av.w * av.w);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
            const float4 wb = *reinterpret_cast<const float4 *>(wat_row);
            c0 = mfma16(wb.x, av.x, c0);
            c1 = mfma16(wb.y, av.y, c1);
            c0 = mfma16(wb.z, av.z, c0);
            c1 = mfma16(wb.w, av.w, c1);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                if (tt < ns) {
                    const float4 wa = *reinterpret_cast<const float4 *>(wst_row + 16 * tt);
                    const float4 xr = *reinterpret_cast<const float4 *>(XS + L.l15 * SR_XLD + 16 * tt + 4 * L.q);
                    c0 = mfma16(wa.x, xr.x, c0);
                    c1 = mfma16(wa.y, xr.y, c1);
                    c0 = mfma16(wa.z, xr.z, c0);
                    c1 = mfma16(wa.w, xr.w, c1);
                }
            }
            float sq = 0.f, mx = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out[r] = c0[r] + c1[r];
                if (j0 + r < S) {
                    sq += out[r] * out[r];
                    mx = fmaxf(mx, fabsf(out[r]));
                }
            }
            const float a2x = __shfl_xor(a2, 16, 64), sqx = __shfl_xor(sq, 16, 64), mxx = __shfl_xor(mx, 16, 64);
            a2 += a2x; sq += sqx; mx = fmaxf(mx, mxx);
            const float a2y = __shfl_xor(a2, 32, 64), sqy = __shfl_xor(sq, 32, 64), mxy = __shfl_xor(mx, 32, 64);
            a2 += a2y; sq += sqy; mx = fmaxf(mx, mxy);
            if (L.q == 0) { RED[(L.wave * 16 + L.l15) * 2] = sq; RED[(L.wave * 16 + L.l15) * 2 + 1] = mx; }
        }
        lds_barrier();
        if (L.wave < ns) {
            float sq = 0.f, mx = 0.f;
            for (int w = 0; w < ns; ++w) { sq += RED[(w * 16 + L.l15) * 2]; mx = fmaxf(mx, RED[(w * 16 + L.l15) * 2 + 1]); }
            const int sc1 = sc + 1;
            const bool term = mx > 10.f;
            const bool trunc = (sc1 >= g.max_step) && !term;
            const bool done = term || trunc;
            if (j0 < S) {
                if (done) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = philox_normal(g.env_seed, (uint64_t)(ep + 1), (uint32_t)row, (uint32_t)(j0 + r));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) if (j0 + r >= S) out[r] = 0.f;
                *reinterpret_cast<float4 *>(XS + L.l15 * SR_XLD + j0) = make_float4(out[0], out[1], out[2], out[3]);
            }
            if (L.wave == 0 && L.q == 0 && valid) {
                const float rew = -(sq / (float)S) - 0.01f * (a2 / (float)A);
                g.o_rewards[(size_t)t * N + row] = g.reward_scale == 1.0f ? rew : rew * g.reward_scale;      // rewards *= reward_scale
                g.o_undones[(size_t)t * N + row] = term ? 0 : 1;                                               // logical_not
                g.o_unmasks[(size_t)t * N + row] = trunc ? 0 : 1;
            }
            sc = done ? 0 : sc1;
            if (done) ep = ep + 1;
        }
        lds_barrier();
    }
    // ---- hand the environment back; the agent's own copy of the final state
    for (int e = L.tid; e < 16 * 64; e += FT) {
        const int i = e >> 6, k = e & 63;
        if (row0 + i < N && k < S) {
            const float x = XS[i * SR_XLD + k];
            g.env_state[(row0 + i) * S + k] = x;
            if (g.o_last_state) g.o_last_state[(row0 + i) * S + k] = x;
        }
    }
    if (ENV == SR_ENV_SYN) {
        if (L.wave == 0 && L.q == 0 && valid) {
            g.step_count[row] = sc;
            g.episode[row] = ep;
        }
    } else if (L.tid < TS && row0 + L.tid < N) {
        g.step_count[prow] = psc;
        g.episode[prow] = pep;
        g.phys[2 * prow] = th;
        g.phys[2 * prow + 1] = thdot;
    }
}

}  // namespace

// ---- shape class of the fused step ---------------------------------------------------------------------------
#ifdef ERL_PROFILE
extern "C" __attribute__((visibility("default"))) int erl_debug_sac_fused_profile(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fprof), sizeof(long long) * (n < 256 ? n : 256));
}
#endif

bool erl_sac_fused_supported(int S, int A, const int *hidden, int n_hidden, int E, int64_t B)
{
    if (n_hidden != 2 || !hidden) return false;
    const int h0 = hidden[0], h1 = hidden[1];
    return S >= 1 && S <= 64 && A >= 1 && A <= 8 && S + A <= 64 && h0 >= 16 && h0 <= FMAXW && h0 % 16 == 0 && h1 >= 16 && h1 <= FMAXW &&
           h1 % 16 == 0 && E >= 1 && E <= FMAXE && B >= 1 && B <= 4096;
}

constexpr int kCritSplit = 4;          // feature slices per decoder in the critic passes that split (see CriticArgs::split)

int64_t erl_sac_fused_ws_floats(int S, int A, int h0, int h1, int E, int64_t B, int64_t Pa, int64_t Pc)
{
    const int64_t tiles = (B + TS - 1) / TS;
    auto r = [](int64_t n) { return (n + 63) / 64 * 64; };
    int64_t f = 0;
    f += 3 * r(B * A) + 2 * r(B);                                       // a_next, act_pg, eps | lp_next, lp_cur
    f += r((int64_t)kCritSplit * E * B) + 2 * r((int64_t)E * B) + r(B);  // qt (also q_pg; up to kCritSplit partial values each), qc, dq | label
    f += r(B * (S + A)) + r(B * h0);                                    // xa, enc
    f += 2 * r((int64_t)E * B * h1) + r((int64_t)E * B * h0);           // H1e, dZ1e | dEncE
    f += r((int64_t)kCritSplit * E * B * h0);                           // dEncP: the slices' shares of dEnc (split training pass)
    f += r(B * 2 * A) * 2 + 2 * r(B * h0) + 2 * r(B * h1);              // Y, dY | H0, G0 | H1, G1
    f += r((int64_t)kCritSplit * E * B * A) + r(B * h1) + r(B * h0);     // dAct | dZ2, dZ1 (actor)
    f += r(Pa) + r(Pc) + r((int64_t)kCritSplit * E * tiles) + r(tiles) + 64;   // gradients, partial sums, alpha0
    f += 2 * 2 * kDwMaxParts;                                           // the squared-norm pieces of the two dw_table launches (doubles)
    f += 2 * r((int64_t)kCritSplit * B * 2 * A);                        // the slices' shares of the actor's head output (launch (1): both passes)
    return f;
}

#define FUSED_KT_DISPATCH(KERNEL_MACRO) FUSED_KT_DISPATCH_D(d, KERNEL_MACRO)
#define FUSED_KT_DISPATCH_D(DD, KERNEL_MACRO)                                                                    \
    switch (wclass((DD).h0) * 3 + wclass((DD).h1)) {                                                            \
        case 0: KERNEL_MACRO(0, 0); break; case 1: KERNEL_MACRO(0, 1); break; case 2: KERNEL_MACRO(0, 2); break; \
        case 3: KERNEL_MACRO(1, 0); break; case 4: KERNEL_MACRO(1, 1); break; case 5: KERNEL_MACRO(1, 2); break; \
        case 6: KERNEL_MACRO(2, 0); break; case 7: KERNEL_MACRO(2, 1); break; default: KERNEL_MACRO(2, 2); break; \
    }

// ActorSAC.get_action for the off-policy rollout (AgentSAC.py:179-185) as ONE launch: the step's actor_fwd kernel on the rollout's N state
// rows (16 per workgroup), nothing kept but the action (lp_scratch: N floats the kernel also writes).  Same Philox keys and the same
// arithmetic per element as the layered form (erl_sac_explore_action_f32's four launches), summed in this kernel's order.
int erl_sac_explore_fused(const float *actor_params, int S, int A, int h0, int h1, const int64_t *aoff, const float *state, int64_t N,
                          const float *noise, uint64_t seed, uint64_t counter, float *action_out, float *state_out, float *lp_scratch,
                          hipStream_t sa)
{
    FusedDims d{};
    d.S = S; d.A = A; d.E = 1; d.h0 = h0; d.h1 = h1; d.B = N;
    d.aW1 = aoff[0]; d.ab1 = aoff[1]; d.aW2 = aoff[2]; d.ab2 = aoff[3]; d.aWh = aoff[4]; d.abh = aoff[5];
    ActorFwdArgs af{};
    af.P = actor_params; af.d = d; af.X = state; af.noise = noise; af.seed = seed; af.counter = counter;
    af.act_t = action_out; af.lp = lp_scratch; af.Xcopy = state_out;
    const dim3 tgrid((unsigned)((N + TS - 1) / TS)), blk(FT);
#define LAUNCH_ACTOR_EXPLORE(K0, K1) hipLaunchKernelGGL((actor_fwd_kernel<K0, K1>), tgrid, blk, 0, sa, af)
    FUSED_KT_DISPATCH(LAUNCH_ACTOR_EXPLORE)
#undef LAUNCH_ACTOR_EXPLORE
    return erl_hip_status(hipGetLastError(), "erl_sac_explore_action_f32 (fused)");
}

// The whole step.  Pointers / scalars as erl_sac_update_f32 (sac.hip), which validates them and dispatches here.
int erl_sac_update_fused(float *actor_params, float *critic_params, float *target_params, float *alpha_log, float *actor_m, float *actor_v,
                         float *critic_m, float *critic_v, float *alpha_m, float *alpha_v, int S, int A, int h0, int h1, int E,
                         const int64_t *aoff, const int64_t *coff, int64_t Pa, int64_t Pc, const float *state, const float *action,
                         const float *reward, const float *undone, const float *unmask, const float *next_state, const float *is_weight,
                         float *td_error_out, int64_t B, const float *eps_next, const float *eps_cur, uint64_t seed, uint64_t counter,
                         float gamma, float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam, float max_norm,
                         int32_t step, float *objs_out, float *workspace, const ErlRingSample *ring, hipStream_t s)
{
    FusedDims d{};
    d.S = S; d.A = A; d.E = E; d.h0 = h0; d.h1 = h1; d.B = B;
    d.aW1 = aoff[0]; d.ab1 = aoff[1]; d.aW2 = aoff[2]; d.ab2 = aoff[3]; d.aWh = aoff[4]; d.abh = aoff[5];
    d.cWe = coff[0]; d.cbe = coff[1]; d.cdec0 = coff[2]; d.dW1 = coff[3]; d.db1 = coff[4]; d.dWo = coff[5]; d.dbo = coff[6]; d.dec = coff[7];
    const int tiles = (int)((B + TS - 1) / TS);
    auto r = [](int64_t n) { return (n + 63) / 64 * 64; };
    float *w = workspace;
    auto take = [&](int64_t n) { float *p = w; w += r(n); return p; };
    float *a_next = take(B * A), *eps_used = take(B * A), *lp_next = take(B), *lp_cur = take(B);
    float *qt = take((int64_t)kCritSplit * E * B), *qc = take((int64_t)E * B), *dq = take((int64_t)E * B), *label = take(B);
    float *xa = take(B * (S + A)), *enc = take(B * h0);
    float *H1e = take((int64_t)E * B * h1), *dZ1e = take((int64_t)E * B * h1), *dEncE = take((int64_t)E * B * h0);
    float *Y = take(B * 2 * A), *dY = take(B * 2 * A), *H0 = take(B * h0), *G0 = take(B * h0), *H1 = take(B * h1), *G1 = take(B * h1);
    float *dAct = take((int64_t)kCritSplit * E * B * A), *dZ2 = take(B * h1), *dZ1 = take(B * h0);
    float *g_actor = take(Pa), *g_critic = take(Pc), *qpart = take((int64_t)kCritSplit * E * tiles), *tdpart = take(tiles), *alpha0 = take(64);
    float *act_pg = take(B * A);                        // (its own buffer: the policy-gradient sample runs next to the critic update)
    double *nparts_c = reinterpret_cast<double *>(take(2 * kDwMaxParts)), *nparts_a = reinterpret_cast<double *>(take(2 * kDwMaxParts));
    float *ypart = take((int64_t)kCritSplit * B * 2 * A);      // the slices' shares of the actor's head output (launch (1))
    float *ypart2 = take((int64_t)kCritSplit * B * 2 * A);     // ... of the policy-gradient sample's forward when it is split too
    float *dEncP = take((int64_t)kCritSplit * E * B * h0);     // the slices' shares of dEnc (launch (3))
    float *q_pg = qt;                                   // reused once its first contents are consumed
    const dim3 tgrid(tiles), cgrid(tiles, E), blk(FT);
    int rc;

    // ---- (1) next action / log-prob (actor on next_state)                                                      (:50-51)
    ActorFwdArgs af{};
    af.P = actor_params; af.d = d; af.X = next_state; af.noise = eps_next; af.seed = seed; af.counter = 2 * counter;
    af.act_t = a_next; af.lp = lp_next; af.alpha_log = alpha_log; af.alpha0 = alpha0;
    if (ring) {
        // the replay sample rides in this launch: the batch pointers are the staging block it fills (sac.hip erl_sac_update_ring_f32)
        af.rg = *ring;
        af.o_state = const_cast<float *>(state); af.o_action = const_cast<float *>(action); af.o_reward = const_cast<float *>(reward);
        af.o_undone = const_cast<float *>(undone); af.o_unmask = const_cast<float *>(unmask); af.o_next = const_cast<float *>(next_state);
    }
    hipStream_t sa = s;                                 // the stream the actor-forward launches go to
#define LAUNCH_ACTOR_FWD(K0, K1) hipLaunchKernelGGL((actor_fwd_kernel<K0, K1>), tgrid, blk, 0, sa, af)
    SacSide *side = sac_side_stream(s);
    // The policy-gradient sample (6) runs on a side stream.  Round 6 forks it BEFORE launch (1) -- it reads the actor and the drawn
    // transitions' state rows (from the ring itself when the sample rides in launch (1): ActorFwdArgs::rg_self), nothing launch (1)
    // writes -- so that it overlaps launches (1) and (2) and is gone by the time the critic's training pass (3) starts: that pass's 256
    // workgroups wait for each other (CriticArgs::qx) and need the chip to themselves.  Only the temperature step behind it waits for
    // launch (1), which parks the old temperature (event `mid`).  ERL_SAC_FORK=2: fork after launch (1) as in rounds 4-5; 0: no fork.
    const char *fk_env = getenv("ERL_SAC_FORK");
    const bool do_fork = side && !(fk_env && atoi(fk_env) == 0);
    static const bool split_on = [] { const char *e = getenv("ERL_SAC_SPLIT"); return !(e && atoi(e) == 0); }();
    static const bool asplit_on = [] { const char *e = getenv("ERL_SAC_SPLIT"); return !(e && atoi(e) == 2); }();      // (2: the critic passes only)
    const bool a_split = split_on && asplit_on && side && side->arrive && h1 == 64 * kCritSplit && tiles * kCritSplit <= 256;
    // round 6: launch (1) and the policy-gradient sample's forward as ONE launch (actor_fwd_pair_kernel; the [256, 256] actor with launch (1)
    // split); only the temperature step is left for the side stream, forked behind that launch
    const char *pr_env = getenv("ERL_SAC_PAIR");
    const bool pair = do_fork && a_split && !(pr_env && atoi(pr_env) == 0) && wclass(h0) == 2 && wclass(h1) == 2 && wclass(h1 / kCritSplit) == 0 &&
                      (int64_t)tiles * (kCritSplit + 1) <= 65535;
    const bool fork_early = do_fork && !pair && !(fk_env && atoi(fk_env) == 2);
    // ERL_SAC_PAIR: 4 (default) as 3 with the sample's forward split over kCritSplit workgroups per tile like launch (1); 3: the temperature step rides the critic's weight-gradient launch (4) -- no side stream at all, nine launches on one
    // queue: the events that forked / joined the side stream cost ~20 us of gaps per step (profiles/r06_sac_pair_ab.txt); 2: a launch of its own on
    // the caller's stream; 1: on the side stream; 0: rounds 4-6's side stream for the sample's forward too
    const int pair_mode = !pair ? 0 : (pr_env ? atoi(pr_env) : 4);
    const bool side_on = do_fork && !(pair && pair_mode >= 2);
    const bool alpha_in_dw = pair && pair_mode >= 3;
    SacJoin joiner{side, s};
    if (fork_early) {
        if ((rc = erl_hip_status(hipEventRecord(side->fork, s), "hipEventRecord(fork)"))) return rc;
        if ((rc = erl_hip_status(hipStreamWaitEvent(side->stream, side->fork, 0), "hipStreamWaitEvent(fork)"))) return rc;
        joiner.armed = true;
    }
    auto pg_args = [&](bool from_ring) {                 // the policy-gradient sample's forward: the actor on `state`, everything kept for the backward pass
        ActorFwdArgs af2 = af;
        af2.X = state; af2.noise = eps_cur; af2.counter = 2 * counter + 1; af2.act_t = act_pg; af2.lp = lp_cur; af2.eps_out = eps_used; af2.Y = Y;
        af2.H0 = H0; af2.G0 = G0; af2.H1 = H1; af2.G1 = G1; af2.alpha0 = nullptr;
        af2.rg = ErlRingSample{};                        // (`state`: the caller's batch, or staged by launch (1), which a late fork waits for)
        if (from_ring && ring) {                         // launch (1) is still staging `state`: read the same rows from the ring
            af2.rg = *ring;
            af2.rg.out_ids0 = af2.rg.out_ids1 = nullptr;
            af2.rg_self = 1;
        }
        return af2;
    };
    if (a_split) {
        // (nothing of this pass is kept for a backward pass: a 256-wide second layer is split over kCritSplit workgroups per tile, the last
        // of them to arrive finishes the head -- ActorFwdArgs::split)
        ActorFwdArgs sp = af;
        sp.split = kCritSplit; sp.h1_full = h1; sp.d.h1 = h1 / kCritSplit; sp.Ypart = ypart; sp.arrive = side->arrive;
        // ERL_SAC_YX=0 keeps the last-arriver meeting of the head's shares (read at every call: the tests compare the two forms in one process)
        const char *yx_env = getenv("ERL_SAC_YX");
        const bool yx_on = side->yx && !(yx_env && atoi(yx_env) == 0) && B <= 4096 && 2 * A <= 16;
        if (yx_on) {
            static const uint32_t ylim = [] { const char *e = getenv("ERL_SAC_QX_SPIN"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : 1u << 22; }();
            if (++side->nonce_y == 0) side->nonce_y = 1;
            sp.yx = side->yx; sp.nonce = side->nonce_y; sp.spin_limit = ylim; sp.fault = erl_fault_word(ERL_FAULT_SAC_Q_EXCHANGE);
        }
        const dim3 sgrid(tiles, kCritSplit);
#define LAUNCH_ACTOR_SPLIT(K0, K1) hipLaunchKernelGGL((actor_fwd_kernel<K0, K1>), sgrid, blk, 0, sa, sp)
        if (pair && pair_mode >= 4) {                    // the sample's forward split over kCritSplit workgroups per tile like launch (1): its own counters and shares
            ActorFwdArgs sp2 = pg_args(true);
            sp2.split = kCritSplit; sp2.h1_full = h1; sp2.d.h1 = h1 / kCritSplit; sp2.Ypart = ypart2; sp2.arrive = side->arrive + 128;
            if (yx_on) { sp2.yx = side->yx + kYxPass; sp2.nonce = sp.nonce; sp2.spin_limit = sp.spin_limit; sp2.fault = sp.fault; }
            hipLaunchKernelGGL((actor_fwd_pair_kernel<2, 0, 0>), dim3(tiles, 2 * kCritSplit), blk, 0, sa, sp, sp2, (int)kCritSplit);
        } else if (pair) {
            const ActorFwdArgs af2 = pg_args(true);
            hipLaunchKernelGGL((actor_fwd_pair_kernel<2, 0, 2>), dim3(tiles, kCritSplit + 1), blk, 0, sa, sp, af2, (int)kCritSplit);
        } else {
            FUSED_KT_DISPATCH_D(sp.d, LAUNCH_ACTOR_SPLIT)
        }
#undef LAUNCH_ACTOR_SPLIT
    } else {
        FUSED_KT_DISPATCH(LAUNCH_ACTOR_FWD)
    }
    // ---- (6) policy-gradient sample (actor on state, kept for the backward pass) and temperature step              (:72-79)
    // FORKED here onto the side stream: they read the actor, `state` and alpha_log only -- the temperature BEFORE its update
    // is already parked in alpha0 by launch (1) -- and run next to the critic update (2)-(5); joined before (7).
    if (fork_early) {
        if ((rc = erl_hip_status(hipEventRecord(side->mid, s), "hipEventRecord(mid)"))) return rc;     // launch (1) is enqueued
        sa = side->stream;
    } else if (side_on) {
        if ((rc = erl_hip_status(hipEventRecord(side->fork, s), "hipEventRecord(fork)"))) return rc;
        if ((rc = erl_hip_status(hipStreamWaitEvent(side->stream, side->fork, 0), "hipStreamWaitEvent(fork)"))) return rc;
        sa = side->stream;
        joiner.armed = true;
    }
    {
        if (!pair) {
            const ActorFwdArgs af2 = pg_args(fork_early);
            ActorFwdArgs keep = af;
            af = af2;
            FUSED_KT_DISPATCH(LAUNCH_ACTOR_FWD)
            af = keep;
        }
        if (fork_early && (rc = erl_hip_status(hipStreamWaitEvent(sa, side->mid, 0), "hipStreamWaitEvent(mid)"))) return rc;   // alpha0 is parked
        const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
        if (!alpha_in_dw)
            hipLaunchKernelGGL(alpha_step_fused_kernel, dim3(1), dim3(256), 0, sa, lp_cur, B, target_entropy, alpha_log, alpha_m, alpha_v, beta1, beta2,
                               eps_adam, max_norm, (float)((double)lr / bc1), (float)sqrt(bc2));
        if (side_on && (rc = erl_hip_status(hipEventRecord(side->join, sa), "hipEventRecord(join)"))) return rc;
        sa = s;
    }
    // ---- (2) target ensemble on (next_state, next_action)                                                     (:52)
    // (the passes whose backward does not need q -- (2) and (7) -- split every 256-wide decoder over kCritSplit workgroups while the
    // launch stays within the chip: 16 tiles x 4 decoders x 4 slices = 256 workgroups at B = 256, each streaming 64 KB of W1 instead of
    // 256 KB through one CU's ~30 GB/s; ERL_SAC_SPLIT=0 turns it off)
    const int split = (split_on && h1 == 64 * kCritSplit && (int64_t)tiles * E * kCritSplit <= 256) ? kCritSplit : 1;
    FusedDims dsl = d;
    dsl.h1 = h1 / split;
    dim3 cg(tiles, E * split);
    CriticArgs ca{};
    ca.P = target_params; ca.d = dsl; ca.split = split; ca.qt_split = 1; ca.Xs = next_state; ca.Xa = a_next; ca.q = qt;
#define LAUNCH_CRITIC(MODE, K0, K1) hipLaunchKernelGGL((critic_tile_kernel<MODE, K0, K1>), cg, blk, 0, s, ca)
#define LAUNCH_CRITIC0(K0, K1) LAUNCH_CRITIC(0, K0, K1)
#define LAUNCH_CRITIC1(K0, K1) LAUNCH_CRITIC(1, K0, K1)
#define LAUNCH_CRITIC2(K0, K1) LAUNCH_CRITIC(2, K0, K1)
    FUSED_KT_DISPATCH_D(dsl, LAUNCH_CRITIC0)
    // ---- (3) critic training pass: labels, loss gradient, backward to the encoder output                      (:53-62)
    // (round 6: split like (2) and (7); its workgroups exchange q inside the launch -- CriticArgs::qx; ERL_SAC_TRAIN_SPLIT=1 selects it)
    // OFF by default: measured at config 3 (B = 256, 4 critics, profiles/r06_sac_train_split_ab.txt) the split launch is 33.2 us against the
    // unsplit 28.9 -- workgroup (0, 0) itself runs 28.7k cycles instead of 40.8k (tools/sac_fused_profile.py: the decoder's forward 6.7k
    // instead of 18.2k, its backward 2.6k instead of 7.4k) but pays 3.4k for the q exchange and 7.1k for the dEnc shares, and 256 mutually
    // waiting workgroups start later and finish more raggedly than 64 independent ones.  ERL_SAC_TRAIN_SPLIT=1 turns it on (read per call).
    const char *ts_env = getenv("ERL_SAC_TRAIN_SPLIT");
    const bool tsplit_on = ts_env ? atoi(ts_env) == 1 : (fork_early || pair);
    const int tsplit = (split > 1 && tsplit_on && side && side->arrive1 && side->qx && B <= 4096 && kCritSplit == kQxSplit) ? split : 1;
    cg = tsplit > 1 ? dim3(tiles, E * tsplit) : cgrid;
    ca.d = tsplit > 1 ? dsl : d; ca.split = tsplit; ca.qt_split = split; ca.h1_full = h1;
    if (tsplit > 1) {
        if (++side->nonce == 0) side->nonce = 1;
        static const uint32_t lim = [] { const char *e = getenv("ERL_SAC_QX_SPIN"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : 1u << 22; }();
        ca.qx = side->qx; ca.nonce = side->nonce; ca.spin_limit = lim; ca.dEncP = dEncP; ca.arrive = side->arrive1;
        ca.fault = erl_fault_word(ERL_FAULT_SAC_Q_EXCHANGE);
    }
    ca.P = critic_params; ca.Xs = state; ca.Xa = action; ca.q = qc;
    ca.qt = qt; ca.reward = reward; ca.undone = undone; ca.unmask = unmask; ca.lp_next = lp_next; ca.is_weight = is_weight; ca.alpha0 = alpha0;
    ca.gamma = gamma; ca.label = label; ca.dq = dq; ca.xa = xa; ca.enc = enc; ca.H1e = H1e; ca.dZ1e = dZ1e; ca.dEncE = dEncE;
    ca.span = erl_span_slot(ERL_SPAN_SAC_CRITIC_TRAIN, (int64_t)cg.x * cg.y);
    FUSED_KT_DISPATCH_D(ca.d, LAUNCH_CRITIC1)
    ca.span = nullptr; ca.h1_full = 0;
    // ---- (4) every critic weight / bias gradient in one launch
    {
        DwArgs dw{};
        dw.B = B;
        dw_add(dw, dEncE, B * h0, E, h0, xa, S + A, g_critic + d.cWe, g_critic + d.cbe);                 // encoder: dEnc = sum_e
        for (int e = 0; e < E; ++e) {
            float *G = g_critic + d.cdec0 + (int64_t)e * d.dec;
            dw_add(dw, dZ1e + (size_t)e * B * h1, 0, 1, h1, enc, h0, G + d.dW1, G + d.db1);
            dw_add(dw, dq + (size_t)e * B, 0, 1, 1, H1e + (size_t)e * B * h1, h1, G + d.dWo, G + d.dbo);
        }
        dw.norm_parts = nparts_c;
        if (alpha_in_dw) {
            const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
            dw.al_lp = lp_cur; dw.al_n = B; dw.al_target_entropy = target_entropy; dw.al_alpha_log = alpha_log; dw.al_m1 = alpha_m; dw.al_m2 = alpha_v;
            dw.al_beta1 = beta1; dw.al_beta2 = beta2; dw.al_eps = eps_adam; dw.al_max_norm = max_norm; dw.al_step_size = (float)((double)lr / bc1);
            dw.al_bc2_sqrt = (float)sqrt(bc2);
        }
        if ((rc = dw_launch(dw, s))) return rc;
        // ---- (5) clip + Adam on the critic from the launch's squared-norm pieces, soft target update in the same launch  (:69-70)
        if ((rc = erl_clip_adam_parts_soft_f32(critic_params, g_critic, critic_m, critic_v, Pc, nparts_c, dw.p[dw.np - 1].tile0 + dw.p[dw.np - 1].ntiles,
                                               step, lr, beta1, beta2, eps_adam, max_norm, target_params, tau, s)))
            return rc;
    }
    // ---- (7) TARGET ensemble on (state, action_pg): q and d(mean q)/d(action); finishes the critic objective    (:82-83)
    if (side_on && (rc = erl_hip_status(hipStreamWaitEvent(s, side->join, 0), "hipStreamWaitEvent(join)"))) return rc;
    joiner.armed = false;
    cg = dim3(tiles, E * split);
    ca.d = dsl; ca.split = split; ca.qt_split = 1;
    ca.P = target_params; ca.Xs = state; ca.Xa = act_pg; ca.q = q_pg;
    ca.dAct = dAct; ca.qpart = qpart; ca.qc = qc; ca.label_in = label; ca.td_out = td_error_out; ca.tdpart = tdpart;
    FUSED_KT_DISPATCH_D(dsl, LAUNCH_CRITIC2)
    // ---- (8) actor backward, (9) its weight gradients, (10) clip + Adam, alpha clamp                            (:80-85)
    {
        ActorBwdArgs ab{};
        ab.P = actor_params; ab.d = d; ab.Y = Y; ab.act_t = act_pg; ab.eps = eps_used; ab.dAct = dAct; ab.alpha_log = alpha_log; ab.G0 = G0;
        ab.G1 = G1; ab.lp_cur = lp_cur; ab.dY = dY; ab.dZ2 = dZ2; ab.dZ1 = dZ1; ab.tdpart = tdpart; ab.qpart = qpart; ab.ntiles = tiles; ab.nsplit = split;
        ab.objs_out = objs_out;
        // (the one heavy layer of this pass, dH0 = W2^T dZ2, is output-split over kCritSplit workgroups per tile like the critic passes above)
        static const bool bsplit_on = [] { const char *e = getenv("ERL_SAC_SPLIT"); return !(e && (atoi(e) == 2 || atoi(e) == 3)); }();   // (3: all but this one)
        const int bsplit = (split_on && bsplit_on && h0 == 64 * kCritSplit && tiles * kCritSplit <= 256) ? kCritSplit : 1;
        ab.split = bsplit; ab.h0_full = h0; ab.d.h0 = h0 / bsplit;
        const dim3 bgrid(tiles, bsplit);
#define LAUNCH_ACTOR_BWD(K0, K1) hipLaunchKernelGGL((actor_bwd_kernel<K0, K1>), bgrid, blk, 0, s, ab)
        FUSED_KT_DISPATCH_D(ab.d, LAUNCH_ACTOR_BWD)
        DwArgs dw{};
        dw.B = B;
        dw_add(dw, dZ1, 0, 1, h0, state, S, g_actor + d.aW1, g_actor + d.ab1);
        dw_add(dw, dZ2, 0, 1, h1, H0, h0, g_actor + d.aW2, g_actor + d.ab2);
        dw_add(dw, dY, 0, 1, 2 * A, H1, h1, g_actor + d.aWh, g_actor + d.abh);
        dw.clamp_alpha_log = alpha_log;        // (one launch less than a kernel of its own: ~5 us of a 230 us step)
        dw.norm_parts = nparts_a;
        if ((rc = dw_launch(dw, s))) return rc;
        if ((rc = erl_clip_adam_parts_soft_f32(actor_params, g_actor, actor_m, actor_v, Pa, nparts_a, dw.p[dw.np - 1].tile0 + dw.p[dw.np - 1].ntiles, step,
                                               lr, beta1, beta2, eps_adam, max_norm, nullptr, 0.f, s)))
            return rc;
    }
    return erl_hip_status(hipGetLastError(), "erl_sac_update_f32(fused)");
}

// ---- the persistent off-policy rollout (sac_rollout_synenv_kernel) ----
extern "C" int erl_sac_rollout_synenv_supported(int S, int A, const int *hidden, int n_hidden, int64_t N)
{
    static const bool on = [] { const char *e = getenv("ERL_SAC_FUSED"); return !(e && atoi(e) == 0); }();
    return (on && erl_sac_fused_supported(S, A, hidden, n_hidden, 1, N < 1 ? 1 : (N > 4096 ? 4096 : N)) && N >= 1 && N <= 4096) ? 1 : 0;
}

int erl_sac_rollout_fused(const float *actor_params, int S, int A, int h0, int h1, const int64_t *aoff, float *env_state, const float *Ws,
                          const float *Wa, float *phys, int32_t *step_count, int32_t *episode, int max_step, uint64_t env_seed, int64_t N, int64_t H,
                          const float *noise, uint64_t seed, uint64_t counter0, float reward_scale, float *out_states, float *out_actions,
                          float *out_rewards, uint8_t *out_undones, uint8_t *out_unmasks, float *out_last_state, hipStream_t stream)
{
    FusedDims d{};
    d.S = S; d.A = A; d.E = 1; d.h0 = h0; d.h1 = h1; d.B = N;
    d.aW1 = aoff[0]; d.ab1 = aoff[1]; d.aW2 = aoff[2]; d.ab2 = aoff[3]; d.aWh = aoff[4]; d.abh = aoff[5];
    SacRolloutArgs g{};
    g.P = actor_params; g.d = d; g.H = (int)H; g.noise = noise; g.seed = seed; g.counter0 = counter0; g.reward_scale = reward_scale;
    g.o_states = out_states; g.o_actions = out_actions; g.o_rewards = out_rewards; g.o_undones = out_undones; g.o_unmasks = out_unmasks;
    g.o_last_state = out_last_state; g.env_state = env_state; g.Ws = Ws; g.Wa = Wa; g.phys = phys; g.step_count = step_count; g.episode = episode;
    g.max_step = max_step; g.env_seed = env_seed;
    const dim3 grid((unsigned)((N + TS - 1) / TS)), blk(FT);
    const size_t lds_bytes = (size_t)SR_FLOATS * sizeof(float);
    static bool attr[2][9] = {};                         // (the dynamic part alone is beyond 64 KB)
#define LAUNCH_SAC_ROLLOUT_E(K0, K1, EV)                                                                                        \
    do {                                                                                                                        \
        if (!attr[EV][K0 * 3 + K1]) {                                                                                           \
            int rc = erl_hip_status(hipFuncSetAttribute((const void *)sac_rollout_synenv_kernel<K0, K1, EV>,                    \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes),            \
                                    "hipFuncSetAttribute(sac_rollout_synenv_kernel)");                                          \
            if (rc) return rc;                                                                                                  \
            attr[EV][K0 * 3 + K1] = true;                                                                                       \
        }                                                                                                                       \
        hipLaunchKernelGGL((sac_rollout_synenv_kernel<K0, K1, EV>), grid, blk, lds_bytes, stream, g);                           \
    } while (0)
#define LAUNCH_SAC_ROLLOUT_SYN(K0, K1) LAUNCH_SAC_ROLLOUT_E(K0, K1, SR_ENV_SYN)
#define LAUNCH_SAC_ROLLOUT_PEN(K0, K1) LAUNCH_SAC_ROLLOUT_E(K0, K1, SR_ENV_PENDULUM)
    if (phys) { FUSED_KT_DISPATCH(LAUNCH_SAC_ROLLOUT_PEN) }
    else { FUSED_KT_DISPATCH(LAUNCH_SAC_ROLLOUT_SYN) }
#undef LAUNCH_SAC_ROLLOUT_SYN
#undef LAUNCH_SAC_ROLLOUT_PEN
#undef LAUNCH_SAC_ROLLOUT_E
    return erl_hip_status(hipGetLastError(), "erl_sac_rollout_synenv_f32");
}
