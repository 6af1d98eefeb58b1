// One-shot peer-to-peer SUM all-reduce of the flat gradient row (the "better" option): 203 KB per minibatch is
// ~1.3 us of xGMI wire time, so the exchange is latency bound and a ring's 2 (P - 1) hops are the wrong shape for it.
// Every rank keeps a STAGE (two halves, used alternately) and a row of FLAGS in uncached device memory that its peers map
// through HIP IPC.  One launch per all-reduce, on the caller's stream, a few workgroups (always resident together):
//
//   1  copy the caller's buffer into my stage half (parity of the sequence number)
//   2  the workgroup that finishes last publishes my sequence number into the flag word "rank" of EVERY peer (remote
//      4-byte system-scope stores: a waiter polls its own memory, never the fabric)
//   3  wait until all `world` words of my own flag row carry this sequence number (bounded; a lost peer is reported
//      through erl_async_fault_count, never a hang)
//   4  out[i] = stage_0[i] + stage_1[i] + ... in RANK ORDER (own stage locally, the others by peer reads over xGMI, all
//      links in parallel): every rank forms bit-identical sums, so the replicas' weights cannot drift.
//
// Stage reuse is safe without a second handshake: a rank passes the wait of sequence s + 1 only after every peer has
// published s + 1, which each peer does after its launch s has finished reading -- so when launch s + 2 overwrites the half
// launch s used, nobody reads it any more.
//
// STATUS: prototype.  Protocol, handle exchange and arithmetic are tested with two ranks that share ONE GPU through IPC
// (tests/test_parallel_gpu.py); it has never run across xGMI (no multi-GPU box was available), so it is opt-in
// (ERL_DP_COLLECTIVE=p2p) and the RCCL all-reduce stays the default exchange.
#include "erl_common.h"

#include <cstring>

namespace {

constexpr int P2P_MAX_WORLD = 8;
constexpr int P2P_WGS = 32, P2P_THREADS = 256;
constexpr size_t P2P_FLAG_BYTES = 4096;              // flag row (one 64-byte line per sender) + the local arrival counter

struct P2PArgs {
    float *buf;
    int64_t count;
    float *stage[P2P_MAX_WORLD];                     // this launch's half of every rank's stage (own one: local pointer)
    uint32_t *flags[P2P_MAX_WORLD];                  // every rank's flag row
    unsigned *arrive;                                // local arrival counter (monotonic)
    unsigned arrive_target;
    uint32_t seq;
    int rank, world;
    uint32_t *fault;
};

__global__ __launch_bounds__(P2P_THREADS) void p2p_allreduce_kernel(P2PArgs g)
{
    __shared__ int s_last;
    const int64_t per = (g.count + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < g.count ? lo + per : g.count;
    float *mine = g.stage[g.rank];
    for (int64_t i = lo + threadIdx.x; i < hi; i += P2P_THREADS) __builtin_nontemporal_store(g.buf[i], mine + i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(g.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old + 1u == g.arrive_target;
    }
    __syncthreads();
    if (s_last && (int)threadIdx.x < g.world) {      // my whole stage half is in memory: tell every rank (myself included)
        __threadfence_system();
        __hip_atomic_store(g.flags[threadIdx.x] + 16 * g.rank, g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if ((int)threadIdx.x < g.world) {
        const uint32_t *f = g.flags[g.rank] + 16 * threadIdx.x;
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - g.seq) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) {
                if (g.fault) __hip_atomic_fetch_add(g.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");     // system scope: drop every cached line of the stages
    for (int64_t i = lo + threadIdx.x; i < hi; i += P2P_THREADS) {
        float s = __builtin_nontemporal_load(g.stage[0] + i);
        for (int r = 1; r < g.world; ++r) s += __builtin_nontemporal_load(g.stage[r] + i);
        g.buf[i] = s;
    }
}

struct P2PComm {
    int rank = 0, world = 1, dev = -1;
    int64_t max_count = 0;
    char *local = nullptr;                           // [flags + counter: P2P_FLAG_BYTES][stage half 0][stage half 1]
    char *peer[P2P_MAX_WORLD] = {};                  // mapped bases (peer[rank] == local)
    bool opened[P2P_MAX_WORLD] = {};
    uint32_t seq = 0;
    unsigned arrive_base = 0;
    bool connected = false;
};

size_t p2p_bytes(int64_t max_count) { return P2P_FLAG_BYTES + 2 * (size_t)((max_count + 63) / 64 * 64) * sizeof(float); }

}  // namespace

// opaque to comm.cpp -------------------------------------------------------------------------------------------------
int erl_p2p_create(int rank, int world, int64_t max_count, void **out, uint8_t *out_handle)
{
    ERL_REQUIRE(out && out_handle && world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world && max_count >= 1,
                "erl_comm_p2p_create: bad argument (world <= %d)", P2P_MAX_WORLD);
    static_assert(sizeof(hipIpcMemHandle_t) == ERL_P2P_HANDLE_BYTES, "hipIpcMemHandle_t size changed");
    P2PComm *c = new P2PComm;
    c->rank = rank; c->world = world; c->max_count = max_count;
    int rc = erl_hip_status(hipGetDevice(&c->dev), "hipGetDevice");
    void *p = nullptr;
    if (!rc) rc = erl_hip_status(hipExtMallocWithFlags(&p, p2p_bytes(max_count), hipDeviceMallocUncached), "hipExtMallocWithFlags(uncached stage)");
    if (!rc) rc = erl_hip_status(hipMemset(p, 0, p2p_bytes(max_count)), "hipMemset(stage)");
    if (!rc) rc = erl_hip_status(hipDeviceSynchronize(), "hipDeviceSynchronize");     // zeroed before any peer can learn the handle
    hipIpcMemHandle_t h;
    if (!rc) rc = erl_hip_status(hipIpcGetMemHandle(&h, p), "hipIpcGetMemHandle");
    if (rc) {
        if (p) (void)hipFree(p);
        delete c;
        return rc;
    }
    c->local = (char *)p;
    c->peer[rank] = c->local;
    memcpy(out_handle, &h, sizeof(h));
    *out = c;
    return ERL_OK;
}

int erl_p2p_connect(void *p2p, const uint8_t *handles)
{
    P2PComm *c = (P2PComm *)p2p;
    ERL_REQUIRE(c && handles, "erl_comm_p2p_connect: NULL argument");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * ERL_P2P_HANDLE_BYTES, sizeof(h));
        void *p = nullptr;
        int rc = erl_hip_status(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
        if (rc) return rc;
        c->peer[r] = (char *)p;
        c->opened[r] = true;
    }
    c->connected = true;
    return ERL_OK;
}

int erl_p2p_allreduce(void *p2p, float *buf, int64_t count, hipStream_t stream)
{
    P2PComm *c = (P2PComm *)p2p;
    ERL_REQUIRE(c && c->connected, "erl_comm_allreduce_sum_f32: peer stages are not connected (erl_comm_p2p_connect)");
    ERL_REQUIRE(count <= c->max_count, "erl_comm_allreduce_sum_f32: %lld floats > the %lld the peer stages were sized for", (long long)count,
                (long long)c->max_count);
    P2PArgs g{};
    g.buf = buf; g.count = count; g.rank = c->rank; g.world = c->world;
    g.seq = ++c->seq;
    const size_t half = (size_t)((c->max_count + 63) / 64 * 64) * sizeof(float);
    for (int r = 0; r < c->world; ++r) {
        g.stage[r] = reinterpret_cast<float *>(c->peer[r] + P2P_FLAG_BYTES + (g.seq & 1u) * half);
        g.flags[r] = reinterpret_cast<uint32_t *>(c->peer[r]);
    }
    g.arrive = reinterpret_cast<unsigned *>(c->local + P2P_FLAG_BYTES - 64);
    int wgs = (int)erl_cdiv(count, 4 * P2P_THREADS);
    wgs = wgs < 1 ? 1 : (wgs > P2P_WGS ? P2P_WGS : wgs);
    c->arrive_base += (unsigned)wgs;                 // wraps with the device counter: equality is all the kernel tests
    g.arrive_target = c->arrive_base;
    g.fault = erl_fault_word();
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(wgs), dim3(P2P_THREADS), 0, stream, g);
    return erl_hip_status(hipGetLastError(), "erl_comm_allreduce_sum_f32(p2p)");
}

void erl_p2p_destroy(void *p2p)
{
    P2PComm *c = (P2PComm *)p2p;
    if (!c) return;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
        if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->local) (void)hipFree(c->local);
    delete c;
}
