// The one exchange step of the path: a SUM all-reduce of the flat actor+critic gradient per minibatch,
// enqueued by RCCL on the SAME stream as the kernels either side of it -- no second stream, no event hand-offs, and
// (through erl_ppo_update_dp_f32) no interpreter between ppo_step, grad_reduce, the collective and clip_adam.
//
// RCCL is bound at run time (dlopen of the librccl.so.1 already mapped by PyTorch when there is one), so the library
// loads and every single-GPU entry point works on a box without RCCL; the comm entry points then fail loudly.
#include <stdlib.h>

#include "erl_common.h"
#include "s3_image.h"
#include "ppo_step_wd.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

namespace {

constexpr int kDefaultTail = 0;      // ERL_FUSED_TAIL default (see erl_ppo_update_dp_f32)
constexpr int kDefaultChains = 1;    // ERL_PPO_CHAINS default (see erl_ppo_update_dp_f32): 1 = one chain of launches, 2 = one per network

// The two-chain update loop's second stream: library-owned, one per (device, caller's stream), non-blocking, with the two events that
// fork it from / join it into the caller's stream.  NULL when the runtime refuses (the loop then stays on one stream).
struct ChainSide {
    int device = -1;
    hipStream_t owner = nullptr, stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
ChainSide g_chain_side[16];
int g_last_chains = 0;               // what the last erl_ppo_update_dp_f32 of this process ran as (erl_ppo_update_chains)
std::mutex g_chain_mu;

ChainSide *chain_side(hipStream_t owner)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_chain_mu);
    for (auto &q : g_chain_side)
        if (q.stream && q.device == dev && q.owner == owner) return &q;
    for (auto &q : g_chain_side) {
        if (q.stream) continue;
        if (hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&q.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            q.stream = nullptr;
            return nullptr;
        }
        q.device = dev;
        q.owner = owner;
        return &q;
    }
    return nullptr;
}

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)          // prefer the copy PyTorch already mapped: one RCCL per process
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.handle)
            for (const char *n : names)
                if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.handle) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
    });
    return r;
}

int rccl_status(ncclResult_t e, const char *what)
{
    if (e == ncclSuccess) return ERL_OK;
    erl_set_error("%s: %s", what, rccl().GetErrorString(e));
    return -(2000 + (int)e);
}

#define ERL_NEED_RCCL(what) ERL_REQUIRE(rccl().ok, what ": librccl.so.1 could not be loaded or lacks the nccl* symbols")

struct Comm {
    ncclComm_t nccl;
    int rank, world;
    void *p2p = nullptr;            // one-shot peer-to-peer exchange (p2p.hip) instead of RCCL
};

} // namespace

extern "C" int erl_comm_unique_id(uint8_t *out_id)
{
    ERL_REQUIRE(out_id, "erl_comm_unique_id: NULL output");
    ERL_NEED_RCCL("erl_comm_unique_id");
    ncclUniqueId id;
    int rc = rccl_status(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    static_assert(sizeof(id) == ERL_COMM_ID_BYTES, "RCCL unique id size changed");
    memcpy(out_id, &id, sizeof(id));
    return ERL_OK;
}

extern "C" int erl_comm_init(const uint8_t *id_bytes, int rank, int world_size, void **out_comm)
{
    ERL_REQUIRE(id_bytes && out_comm, "erl_comm_init: NULL argument");
    ERL_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "erl_comm_init: rank %d not in [0, %d)", rank, world_size);
    ERL_NEED_RCCL("erl_comm_init");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    int rc = rccl_status(rccl().CommInitRank(&c, world_size, id, rank), "ncclCommInitRank");   // binds hipGetDevice()'s GPU
    if (rc) return rc;
    *out_comm = new Comm{c, rank, world_size};
    return ERL_OK;
}

extern "C" int erl_comm_destroy(void *comm)
{
    if (!comm) return ERL_OK;
    Comm *c = (Comm *)comm;
    if (c->p2p) {
        erl_p2p_destroy(c->p2p);
        delete c;
        return ERL_OK;
    }
    int rc = rccl().ok ? rccl_status(rccl().CommDestroy(c->nccl), "ncclCommDestroy") : ERL_OK;
    delete c;
    return rc;
}

extern "C" int erl_comm_world_size(void *comm) { return comm ? ((Comm *)comm)->world : 1; }

extern "C" int erl_comm_allreduce_sum_f32(void *comm, float *buf, int64_t count, void *stream)
{
    ERL_REQUIRE(comm && buf && count >= 0, "erl_comm_allreduce_sum_f32: bad argument");
    if (count == 0) return ERL_OK;
    Comm *c = (Comm *)comm;
    if (c->p2p) {                            // the fused tail's kernel with ONE "slab": push, flag, wait, rank-ordered sum
        ErlExchange ex;
        int rc = erl_p2p_next(c->p2p, &ex);
        if (rc) return rc;
        return erl_launch_reduce_exchange_f32(buf, 1, count, buf, nullptr, nullptr, 0, 1.f, false, &ex, (hipStream_t)stream);
    }
    return rccl_status(rccl().AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, c->nccl, (hipStream_t)stream),
                       "ncclAllReduce");
}

extern "C" int erl_comm_allreduce_sum_f64(void *comm, double *buf, int64_t count, void *stream)
{
    ERL_REQUIRE(comm && buf && count >= 0, "erl_comm_allreduce_sum_f64: bad argument");
    if (count == 0) return ERL_OK;
    Comm *c = (Comm *)comm;
    if (c->p2p) {
        ErlExchange ex;
        int rc = erl_p2p_next(c->p2p, &ex);
        if (rc) return rc;
        return erl_launch_exchange_f64(buf, count, &ex, (hipStream_t)stream);
    }
    return rccl_status(rccl().AllReduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, c->nccl, (hipStream_t)stream),
                       "ncclAllReduce(f64)");
}

uint32_t *erl_comm_poison_word(void *comm) { return comm && ((Comm *)comm)->p2p ? erl_p2p_poison_word(((Comm *)comm)->p2p) : nullptr; }

extern "C" int erl_comm_kind(void *comm) { return comm ? (((Comm *)comm)->p2p ? ERL_COMM_KIND_P2P : ERL_COMM_KIND_RCCL) : -1; }

// launch 1 of the data-parallel optimiser tail on its own (the update loop below calls the same code): slab reduction +
// exchange + partial norms for erl_clip_adam_partials_f32.  RCCL communicators take three launches for it.
extern "C" int erl_comm_reduce_exchange_f32(void *comm, const float *slabs, int n_slabs, int64_t stride, float *flat_grad,
                                            const int64_t *group_off, const int64_t *group_len, int n_groups, float grad_scale, void *stream)
{
    Comm *c = (Comm *)comm;
    hipStream_t st = (hipStream_t)stream;
    if (!c) return erl_launch_reduce_exchange_f32(slabs, n_slabs, stride, flat_grad, group_off, group_len, n_groups, grad_scale, true, nullptr, st);
    if (c->p2p) {
        ErlExchange ex;
        int rc = erl_p2p_next(c->p2p, &ex);
        if (rc) return rc;
        return erl_launch_reduce_exchange_f32(slabs, n_slabs, stride, flat_grad, group_off, group_len, n_groups, grad_scale, true, &ex, st);
    }
    int rc = erl_launch_reduce_exchange_f32(slabs, n_slabs, stride, flat_grad, nullptr, nullptr, 0, 1.f, false, nullptr, st);
    if (!rc) rc = erl_comm_allreduce_sum_f32(comm, flat_grad, stride, stream);
    if (!rc) rc = erl_grad_sq_partials_f32(flat_grad, stride, group_off, group_len, n_groups, grad_scale, stream);
    return rc;
}

// ---- one-shot peer-to-peer communicator (p2p.hip): same handle type, same erl_comm_allreduce_sum_f32 / erl_ppo_update_dp_f32
extern "C" int erl_comm_p2p_create(int rank, int world_size, int64_t max_count, void **out_comm, uint8_t *out_handle)
{
    ERL_REQUIRE(out_comm && out_handle, "erl_comm_p2p_create: NULL argument");
    void *p = nullptr;
    int rc = erl_p2p_create(rank, world_size, max_count, &p, out_handle);
    if (rc) return rc;
    Comm *c = new Comm{nullptr, rank, world_size};
    c->p2p = p;
    *out_comm = c;
    return ERL_OK;
}

extern "C" int erl_comm_p2p_connect(void *comm, const uint8_t *handles)
{
    ERL_REQUIRE(comm && ((Comm *)comm)->p2p, "erl_comm_p2p_connect: not a peer-to-peer communicator");
    return erl_p2p_connect(((Comm *)comm)->p2p, handles);
}

extern "C" int erl_comm_p2p_set_spin(void *comm, uint32_t spins)
{
    ERL_REQUIRE(comm && ((Comm *)comm)->p2p, "erl_comm_p2p_set_spin: not a peer-to-peer communicator");
    erl_p2p_set_spin(((Comm *)comm)->p2p, spins);
    return ERL_OK;
}

// 1 / 2: the form the last update loop of this process took (0: none yet) -- measurement and logging
extern "C" int erl_ppo_update_chains(void) { return g_last_chains; }

extern "C" int erl_ppo_update_dp_f32(float *flat_params, float *exp_avg, float *exp_avg_sq, const float *act_avg, const float *act_std,
                                     const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A, const float *states,
                                     const float *actions, const uint8_t *unmasks, const float *logprobs, const float *advantages,
                                     const float *reward_sums, int64_t H, int64_t N, const int64_t *ids, int64_t B,
                                     int update_times, float ratio_clip, float lambda_entropy, int objective, float *slabs, float *grads,
                                     int32_t first_step, float lr, float beta1, float beta2, float eps, float max_norm, double *adv_stats,
                                     const double *adv_partials, int n_partials, void *comm, void *stream)
{
    ERL_REQUIRE(flat_params && exp_avg && exp_avg_sq && ids && slabs && grads, "erl_ppo_update_dp_f32: NULL tensor");
    ERL_REQUIRE(update_times >= 1 && first_step >= 1 && B >= 1, "erl_ppo_update_dp_f32: bad argument");
    const int64_t Pa = erl_mlp_param_count(S, h1, h2, A, 1), Pc = erl_mlp_param_count(S, h1, h2, 1, 0);
    ERL_REQUIRE(Pa > 0 && Pc > 0, "erl_ppo_update_dp_f32: unsupported dims S=%d net=[%d,%d] A=%d", S, h1, h2, A);
    const int64_t stride = erl_ppo_slab_stride(S, h1, h2, A);       // Pa + Pc + 4 rounded up to whole 128-byte lines
    const int n_slabs = erl_ppo_num_slabs(B);
    const int64_t off[2] = {0, Pa}, len[2] = {Pa, Pc};
    const int world = erl_comm_world_size(comm);
    const float grad_scale = 1.0f / (float)world;          // SUM over ranks -> mean, folded into the optimiser
    // The optimiser tail of a single process (nothing sits between the reduction and the optimiser): ERL_FUSED_TAIL selects
    // 0 = grad_reduce + clip_adam (two launches), 1 = one launch whose last-arriving workgroup applies Adam alone (measured
    // slower: profiles/r02_tail_bench.txt), 2 = one launch in which every workgroup waits for the norm and updates its own
    // elements (needs the whole grid resident: erl_reduce_clip_adam_grid_ok).  Default: see kDefaultTail.
    const int tail_env = [] { const char *e = getenv("ERL_FUSED_TAIL"); return e ? atoi(e) : kDefaultTail; }();     // (read per call: A/B in one process)
    // 3 (round 5) = tail_fused_kernel (grad_tail.hip): the two-launch tail's arithmetic in one launch -- partial norms as their own flags,
    // clip + Adam from registers, weight images refreshed: the split-arithmetic minibatch kernel keeps its images
    const bool fused3 = !comm && tail_env == 3 && erl_tail_fused_ok(stride);
    const int tail = (comm || tail_env == 3) ? 0 : (tail_env == 2 && !erl_reduce_clip_adam_grid_ok(stride) ? 0 : tail_env);
    // split-arithmetic minibatch kernel: its W2 images are built here once and kept current by clip + Adam (two-launch tail only)
    S3Images images{}, *im = nullptr;
    ERL_REQUIRE(!adv_partials || (adv_stats && n_partials >= 1), "erl_ppo_update_dp_f32: adv_partials needs adv_stats and n_partials");
    bool folded = adv_partials == nullptr;
    if (!tail && erl_ppo_arith_for_call(S, h1, h2, A, (objective >> 8) & 3) == ERL_PPO_ARITH_SPLIT) {
        // (the (128 | 64, h2) kernels also get the per-sample records: s3_image.h)
        const S3AuxSrc ax{actions, logprobs, advantages, reward_sums, unmasks, A, H * N};
        int rc = erl_s3_images_build(flat_params, S, h1, h2, A, &images, adv_partials, n_partials, H, N, adv_stats, (hipStream_t)stream,
                                     erl_ppo_wd_supported(S, h1, h2, A) ? nullptr : &ax);
        if (rc) return rc;
        im = &images;
        folded = true;
    }
    if (!folded) {           // no image launch to ride: the fold is a launch of its own
        int rc = erl_adv_stats_fold_f32(adv_partials, n_partials, H, N, adv_stats, stream);
        if (rc) return rc;
    }
    // ---- two chains (round 6).  The actor's and the critic's minibatches never meet: each network has its own gradient, its own clip norm
    // and its own Adam step (optimizer_backward is called per optimiser: elegantrl/agents/AgentPPO.py:196-204, AgentBase.py:239-248), so
    // minibatch kernel -> slab reduction -> clip + Adam is one dependent chain PER NETWORK.  ERL_PPO_CHAINS=2 runs them as two chains of
    // half-chip launches on two streams: one chain's launch-bound tail and its gradient write-out overlap the other chain's arithmetic.
    // Same kernels, same associations: the parameters are bit-identical to the one-chain loop's.  Single process, split-arithmetic
    // (128 | 64, h2) kernels with their images, a device that keeps workgroup map 0 (on a slow-fetch device the full-chip launch under
    // map 2 stays: two half launches would put both code paths through every instruction cache again).
    const int chains_env = [] { const char *e = getenv("ERL_PPO_CHAINS"); return e ? atoi(e) : kDefaultChains; }();
    if (chains_env == 2 && !comm && !tail && !fused3 && im && !erl_ppo_wd_supported(S, h1, h2, A) && erl_k6_wg_map_choice(0) == 0) {
        hipStream_t s0 = (hipStream_t)stream;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        ChainSide *side = (hipStreamIsCapturing(s0, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) ? chain_side(s0) : nullptr;
        (void)hipGetLastError();
        if (side) {
            int rc;
            g_last_chains = 2;
            if ((rc = erl_hip_status(hipEventRecord(side->fork, s0), "hipEventRecord(fork)"))) return rc;
            if ((rc = erl_hip_status(hipStreamWaitEvent(side->stream, side->fork, 0), "hipStreamWaitEvent(fork)"))) return rc;
            const hipStream_t cs[2] = {s0, side->stream};
            for (int k = 0; k < update_times && !rc; ++k) {
                float *g = grads + (size_t)k * stride;
                for (int c = 0; c < 2 && !rc; ++c) {
                    rc = erl_ppo_step_images_f32(flat_params, flat_params + Pa, act_avg, act_std, cri_avg, cri_std, S, h1, h2, A, states, actions, unmasks,
                                                 logprobs, advantages, reward_sums, H, N, ids + (size_t)k * B, B, ratio_clip, lambda_entropy,
                                                 1.0f / (float)B, objective, slabs, n_slabs, im, adv_stats, nullptr, cs[c], c);
                    if (!rc) rc = erl_launch_reduce_group_f32(slabs, n_slabs, stride, g, off, len, 2, c, grad_scale, cs[c]);
                    if (!rc) rc = erl_clip_adam_partials_group_f32(flat_params, g, exp_avg, exp_avg_sq, stride, off, len, 2, c, first_step + k, lr, beta1,
                                                                   beta2, eps, max_norm, grad_scale, im, cs[c]);
                }
            }
            // every exit joins the side stream back (an error above leaves both streams consistent: whatever was enqueued runs)
            (void)hipEventRecord(side->join, side->stream);
            (void)hipStreamWaitEvent(s0, side->join, 0);
            return rc;
        }
    }
    g_last_chains = 1;
    erl_k6_touch_next_launch();       // the loop's first launch finds the kernel's code in no cache (ppo_step.h, k6_code_touch)
    for (int k = 0; k < update_times; ++k) {
        float *g = grads + (size_t)k * stride;
        int rc = erl_ppo_step_images_f32(flat_params, flat_params + Pa, act_avg, act_std, cri_avg, cri_std, S, h1, h2, A, states, actions,
                                         unmasks, logprobs, advantages, reward_sums, H, N, ids + (size_t)k * B, B, ratio_clip,
                                         lambda_entropy, 1.0f / (float)B, objective, slabs, n_slabs, im, adv_stats,
                                         k + 1 < update_times ? ids + (size_t)(k + 1) * B : nullptr, stream);
        if (rc) return rc;
        if (tail) {
            rc = (tail == 2 ? erl_reduce_clip_adam_grid_f32 : erl_reduce_clip_adam_f32)(slabs, n_slabs, stride, g, flat_params, exp_avg, exp_avg_sq,
                                                                                        off, len, 2, first_step + k, lr, beta1, beta2, eps,
                                                                                        max_norm, grad_scale, stream);
            if (rc) return rc;
            continue;
        }
        if (fused3) {
            if ((rc = erl_tail_fused_f32(slabs, n_slabs, stride, g, off, len, 2, grad_scale, flat_params, exp_avg, exp_avg_sq, first_step + k, lr, beta1,
                                         beta2, eps, max_norm, im, (hipStream_t)stream)))
                return rc;
            continue;
        }
        // launch 1: slab reduction [+ the exchange, inside the same kernel on a peer-to-peer communicator] + partial norms;
        // launch 2: clip + Adam from the partial norms.  Gradient + the 3 logged objectives travel in one row.
        if ((rc = erl_comm_reduce_exchange_f32(comm, slabs, n_slabs, stride, g, off, len, 2, grad_scale, stream))) return rc;
        if ((rc = erl_clip_adam_partials_images_f32(flat_params, g, exp_avg, exp_avg_sq, stride, off, len, 2, first_step + k, lr, beta1, beta2,
                                                    eps, max_norm, grad_scale, im, erl_comm_poison_word(comm), stream)))
            return rc;
    }
    return ERL_OK;
}
