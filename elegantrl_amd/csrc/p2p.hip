// One-shot peer-to-peer exchange of the flat gradient row (the "better" option): 203 KB per minibatch is a few
// microseconds of xGMI wire time, so the exchange is latency bound and a ring's 2 (P - 1) hops are the wrong shape for it.
// This file owns the memory: every rank keeps a FLAG table [sender][workgroup] and a STAGE of two halves x `world` rows in
// uncached device memory that its peers map through HIP IPC.  The kernels that use it live in grad_tail.hip
// (reduce_exchange_kernel): a rank PUSHES its reduced slice into its row of every peer's stage (posted remote stores, one-way
// latency; no round trips over the fabric), raises a per-workgroup sequence flag on every peer, and each rank then polls and
// sums its OWN memory in rank order.  The protocol, its reuse argument and its forward-progress argument are in the header
// of grad_tail.hip.
//
// Whether the route is used at all is decided at communicator creation by a self-test against RCCL (parallel.py): a route
// that cannot be validated on the machine it runs on is never selected.
#include "erl_common.h"

#include <cstdlib>
#include <cstring>

namespace {

constexpr size_t P2P_ALIGN = 4096;

struct P2PComm {
    int rank = 0, world = 1, dev = -1;
    int64_t max_count = 0;               // floats per exchange the rows are sized for
    int64_t row_bytes = 0, half_bytes = 0, flag_bytes = 0;
    int nblk_max = 0;
    char *local = nullptr;               // [flags: world x nblk_max words][half 0: world rows][half 1: world rows]
    char *peer[ERL_P2P_MAX_WORLD] = {};  // mapped bases (peer[rank] == local)
    bool opened[ERL_P2P_MAX_WORLD] = {};
    uint32_t seq = 0;
    uint32_t *poison = nullptr;          // device word (ordinary memory): a wait of the current update loop timed out
    uint32_t spin = 1u << 24;            // polls before a wait gives up (~1 us each: tens of seconds; ERL_P2P_SPIN / erl_comm_p2p_set_spin)
    bool connected = false;
};

size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

P2PComm *g_live[64] = {};            // live communicators of this process (poison words are cleared through it)

}  // namespace

int erl_p2p_create(int rank, int world, int64_t max_count, void **out, uint8_t *out_handle)
{
    ERL_REQUIRE(out && out_handle && world >= 1 && world <= ERL_P2P_MAX_WORLD && rank >= 0 && rank < world && max_count >= 1 &&
                    max_count <= (1LL << 26),
                "erl_comm_p2p_create: bad argument (world <= %d)", ERL_P2P_MAX_WORLD);
    static_assert(sizeof(hipIpcMemHandle_t) == ERL_P2P_HANDLE_BYTES, "hipIpcMemHandle_t size changed");
    P2PComm *c = new P2PComm;
    c->rank = rank; c->world = world; c->max_count = max_count;
    if (const char *e = getenv("ERL_P2P_SPIN"))
        if (atol(e) > 0) c->spin = (uint32_t)atol(e);
    c->nblk_max = (int)erl_cdiv(max_count, 256);
    c->row_bytes = (int64_t)round_up((size_t)c->nblk_max * 256 * sizeof(float), 256);
    c->half_bytes = (int64_t)world * c->row_bytes;
    // [flags: world x nblk_max words][poisoned: world words -- rank r's word, raised on EVERY rank when a wait of rank r timed out]
    c->flag_bytes = (int64_t)round_up(((size_t)world * c->nblk_max + (size_t)world) * sizeof(uint32_t), P2P_ALIGN);
    const size_t bytes = (size_t)c->flag_bytes + 2 * (size_t)c->half_bytes;
    int rc = erl_hip_status(hipGetDevice(&c->dev), "hipGetDevice");
    void *p = nullptr;
    if (!rc) rc = erl_hip_status(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached), "hipExtMallocWithFlags(uncached stage)");
    if (!rc) rc = erl_hip_status(hipMemset(p, 0, bytes), "hipMemset(stage)");
    if (!rc) rc = erl_hip_status(hipDeviceSynchronize(), "hipDeviceSynchronize");     // zeroed before any peer can learn the handle
    hipIpcMemHandle_t h;
    if (!rc) rc = erl_hip_status(hipIpcGetMemHandle(&h, p), "hipIpcGetMemHandle");
    void *pw = nullptr;
    if (!rc) rc = erl_hip_status(hipMalloc(&pw, 256), "hipMalloc(poison word)");
    if (!rc) rc = erl_hip_status(hipMemset(pw, 0, 256), "hipMemset(poison word)");
    if (!rc) rc = erl_hip_status(hipDeviceSynchronize(), "hipDeviceSynchronize");     // (non-blocking streams are not ordered behind that memset)
    if (rc) {
        if (p) (void)hipFree(p);
        if (pw) (void)hipFree(pw);
        delete c;
        return rc;
    }
    c->poison = (uint32_t *)pw;
    bool listed = false;
    for (auto &g : g_live)
        if (!g) { g = c; listed = true; break; }
    if (!listed) {      // a communicator that is not listed would keep its poison word after a reported fault: refuse it
        (void)hipFree(p);
        (void)hipFree(pw);
        delete c;
        ERL_REQUIRE(false, "erl_comm_p2p_create: more than %d live peer-to-peer communicators in one process", (int)(sizeof(g_live) / sizeof(g_live[0])));
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

int erl_p2p_next(void *p2p, ErlExchange *ex)
{
    P2PComm *c = (P2PComm *)p2p;
    ERL_REQUIRE(c && c->connected && ex, "peer-to-peer exchange: the peer stages are not connected (erl_comm_p2p_connect)");
    *ex = ErlExchange{};
    for (int r = 0; r < c->world; ++r) {
        ex->flags[r] = reinterpret_cast<uint32_t *>(c->peer[r]);
        ex->stage[r] = c->peer[r] + c->flag_bytes;
    }
    ex->half_bytes = c->half_bytes;
    ex->row_bytes = c->row_bytes;
    ex->nblk_max = c->nblk_max;
    ex->seq = ++c->seq;
    ex->spin_limit = c->spin;
    ex->rank = c->rank;
    ex->world = c->world;
    ex->fault = erl_fault_word(ERL_FAULT_P2P_EXCHANGE);
    ex->poison = c->poison;
    return ERL_OK;
}

void erl_p2p_set_spin(void *p2p, uint32_t spins)
{
    if (p2p) ((P2PComm *)p2p)->spin = spins ? spins : (1u << 24);
}

uint32_t *erl_p2p_poison_word(void *p2p) { return p2p ? ((P2PComm *)p2p)->poison : nullptr; }

void erl_p2p_clear_poison_all()
{
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto c : g_live)
        if (c && c->poison) {
            (void)hipSetDevice(c->dev);
            (void)hipDeviceSynchronize();
            (void)hipMemset(c->poison, 0, 4);
            // ... and the words the peers raised in MY table (theirs are cleared by their own report)
            if (c->local) (void)hipMemset(c->local + (size_t)c->world * c->nblk_max * sizeof(uint32_t), 0, (size_t)c->world * sizeof(uint32_t));
            (void)hipDeviceSynchronize();
        }
    if (cur >= 0) (void)hipSetDevice(cur);
}

void erl_p2p_destroy(void *p2p)
{
    P2PComm *c = (P2PComm *)p2p;
    if (!c) return;
    for (auto &g : g_live)
        if (g == c) g = nullptr;
    (void)hipDeviceSynchronize();
    if (c->poison) (void)hipFree(c->poison);
    for (int r = 0; r < c->world; ++r)
        if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->local) (void)hipFree(c->local);
    delete c;
}
