// What would one PPO minibatch's data movement cost INSIDE a persistent kernel?  256 workgroups (one per CU: 157 KB of LDS each,
// like the one-wave K6) loop over `rounds` minibatches of
//   (a) read the 200 KB of weights (written by other workgroups in the previous round),
//   (b) write a 104 KB slab per workgroup,                          -- grid barrier 1
//   (c) reduce one 256-element slice over the 256 slabs, publish the slice's partial norms,   -- grid barrier 2
//   (d) read every partial, update the slice's weights, publish them.                         -- grid barrier 3
// with the cross-workgroup data in cached or fine-grained memory and plain / nt / sc0 sc1 accesses, and flag barriers without
// cache maintenance.  Every round's reduction is checked against its closed form inside the kernel, so a stale line anywhere
// shows up as a mismatch count.  Prints us per round and the phase breakdown of workgroup 0 (s_memtime ticks, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/persist_probe tools/persist_probe.hip && tools/bin/persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NWG = 256, NT = 256, SLICE = 256, NP = NWG * SLICE;     // 65536 "parameters" (real: 50833)
constexpr int kSpin = 1 << 22;

// cache policy bits of a gfx942/950 buffer instruction's aux operand
constexpr int SC0 = 1, NTB = 2, SC1 = 16;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
template <int AUX> __device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX));
}
template <int AUX> __device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), r, byte_off, 0, AUX);
}
__device__ __forceinline__ void st_flag(uint32_t *p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t ld_flag(const uint32_t *p)
{
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ uint64_t now()
{
    uint64_t t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// every workgroup publishes seq in its own word and waits for everybody's; no cache maintenance: the data crossing it is
// written / read with the access flavour under test
template <int FSTRIDE>
__device__ __forceinline__ bool grid_barrier(uint32_t *flags, uint32_t seq, uint32_t *fault)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) st_flag(flags + FSTRIDE * blockIdx.x, seq);
    bool ok = true;
    {
        const uint32_t *f = flags + FSTRIDE * threadIdx.x;       // NT == NWG: thread t watches workgroup t
        int spin = 0;
        while ((int32_t)(ld_flag(f) - seq) < 0) {
            if (++spin > kSpin) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (!ok) atomicAdd(fault, 1u);
    return __syncthreads_and(ok);
}

struct Args {
    float *slabs;        // [NWG][NP]
    float *weights;      // [NP]
    double *partials;    // [NWG][4]
    uint32_t *flags;
    uint32_t *fault;
    uint64_t *stamps;    // [8] phase sums of workgroup 0
    uint32_t *mismatch;
    int rounds;
    uint32_t seq0;
};

template <int ST, int LD, int FSTRIDE>
__global__ __launch_bounds__(NT) void probe(Args a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wg = blockIdx.x;
    const auto rs = rsrc(a.slabs + (size_t)0), rw = rsrc(a.weights), rp = rsrc(a.partials);
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t seq = a.seq0;
    float m1 = 0.f;                                       // the slice's optimiser state stays in registers across rounds
    uint32_t bad = 0;
    for (int r = 0; r < a.rounds; ++r) {
        uint64_t t0 = now();
        // (a) all weights -> LDS (64 float4 per thread)
        {
            f32x4 w[NP / 4 / NT];
#pragma unroll
            for (int i = 0; i < NP / 4 / NT; ++i) w[i] = ld4<LD>(rw, 16 * (tid + NT * i));
#pragma unroll
            for (int i = 0; i < NP / 4 / NT; ++i) {
                if (i < 32) *reinterpret_cast<f32x4 *>(lds + 4 * (tid + NT * i)) = w[i];
                // weights of round r: value r for every element (round 0: 0)
                if (w[i].x != (float)r || w[i].w != (float)r) ++bad;
            }
        }
        uint64_t t1 = now();
        // (b) my slab: element e of workgroup wg in round r = (wg + e + r) & 255   (26 float4 x 4 = 104 KB: 6656 float4)
        {
            float *mine = a.slabs + (size_t)wg * NP;
            const auto rm = rsrc(mine);
#pragma unroll 8
            for (int i = 0; i < 26; ++i) {
                const int q = tid + NT * i, e = 4 * q;
                f32x4 v = {(float)((wg + e + r) & 255), (float)((wg + e + 1 + r) & 255), (float)((wg + e + 2 + r) & 255), (float)((wg + e + 3 + r) & 255)};
                st4<ST>(rm, 16 * q, v);
            }
        }
        uint64_t t2 = now();
        if (!grid_barrier<FSTRIDE>(a.flags, ++seq, a.fault)) return;
        uint64_t t3 = now();
        // (c) slice wg: elements [256 wg, 256 wg + 256) = 64 float4 columns; thread (q = tid & 63, sg = tid >> 6) sums slabs 64 sg .. + 63
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bool live = (SLICE * wg) / 4 + 63 < 6656;          // slices inside the 104 KB that is actually written
        if (live) {
            const int q = tid & 63, sg = tid >> 6;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 x[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const int s = 64 * sg + 32 * h + j;
                    x[j] = ld4<LD>(rsrc(a.slabs + (size_t)s * NP), 16 * (SLICE / 4 * wg + q));
                }
#pragma unroll
                for (int j = 0; j < 32; ++j) acc += x[j];
            }
            __syncthreads();
            *reinterpret_cast<f32x4 *>(lds + 4 * tid) = acc;
            __syncthreads();
            if (tid < 64) {
                acc = *reinterpret_cast<f32x4 *>(lds + 4 * tid);
                for (int g = 1; g < 4; ++g) acc += *reinterpret_cast<f32x4 *>(lds + 4 * (tid + 64 * g));
                // closed form: sum over wg' of ((wg' + e + r) & 255) = 0 + 1 + ... + 255 = 32640 for every element
                if (acc.x != 32640.f || acc.y != 32640.f || acc.z != 32640.f || acc.w != 32640.f) ++bad;
                double sq = (double)acc.x * acc.x + (double)acc.y * acc.y + (double)acc.z * acc.z + (double)acc.w * acc.w;
                for (int o = 32; o; o >>= 1) sq += __shfl_xor(sq, o, 64);
                if (tid == 0) {
                    double *p = a.partials + 4 * wg;
                    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(sq + (double)r) : "memory");
                }
            }
        }
        uint64_t t4 = now();
        if (!grid_barrier<FSTRIDE>(a.flags, ++seq, a.fault)) return;
        uint64_t t5 = now();
        // (d) every partial (thread t: workgroup t's), fixed-order sum, update my slice, publish
        {
            double pv = 0.0;
            {
                const double *p = a.partials + 4 * tid;
                asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(pv) : "v"(p) : "memory");
            }
            const bool tl = (SLICE * tid) / 4 + 63 < 6656;
            if (tl && pv != 64.0 * 4.0 * 32640.0 * 32640.0 + (double)r) ++bad;
            double *ld = reinterpret_cast<double *>(lds);
            __syncthreads();
            ld[tid] = tl ? pv : 0.0;
            __syncthreads();
            double tot = 0.0;
            for (int i = 0; i < NWG; i += 4) tot += (ld[i] + ld[i + 1]) + (ld[i + 2] + ld[i + 3]);
            m1 = 0.9f * m1 + (float)(tot * 1e-30);
            if (tid < 64) {
                const float v = (float)(r + 1) + m1 * 0.f;
                st4<SC0 | SC1>(rw, 16 * (SLICE / 4 * wg + tid), f32x4{v, v, v, v});
            }
        }
        uint64_t t6 = now();
        if (!grid_barrier<FSTRIDE>(a.flags, ++seq, a.fault)) return;
        uint64_t t7 = now();
        ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += t6 - t5; ph[6] += t7 - t6;
    }
    if (bad) atomicAdd(a.mismatch, bad);
    if (wg == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) a.stamps[i] = ph[i];
}

template <int ST, int LD, int FSTRIDE>
void run(const char *name, bool uncached, int rounds)
{
    Args a{};
    auto alloc = [&](void **p, size_t n) {
        if (uncached) CK(hipExtMallocWithFlags(p, n, hipDeviceMallocUncached));
        else CK(hipMalloc(p, n));
        CK(hipMemset(*p, 0, n));
    };
    alloc((void **)&a.slabs, (size_t)NWG * NP * 4);
    alloc((void **)&a.weights, (size_t)NP * 4);
    alloc((void **)&a.partials, (size_t)NWG * 4 * 8);
    CK(hipExtMallocWithFlags((void **)&a.flags, 64 * NWG * 4, hipDeviceMallocUncached));
    CK(hipMemset(a.flags, 0, 64 * NWG * 4));
    CK(hipMalloc((void **)&a.fault, 4)); CK(hipMemset(a.fault, 0, 4));
    CK(hipMalloc((void **)&a.mismatch, 4)); CK(hipMemset(a.mismatch, 0, 4));
    CK(hipMalloc((void **)&a.stamps, 64)); CK(hipMemset(a.stamps, 0, 64));
    a.rounds = rounds;
    const size_t ldsb = 157 * 1024;
    CK(hipFuncSetAttribute((const void *)probe<ST, LD, FSTRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    uint32_t seq = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(a.weights, 0, (size_t)NP * 4));
        CK(hipDeviceSynchronize());
        a.seq0 = seq;
        seq += 3 * rounds;
        void *args[] = {&a};
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel((const void *)probe<ST, LD, FSTRIDE>, dim3(NWG), dim3(NT), args, (unsigned)ldsb, 0));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    uint32_t fault, mism;
    uint64_t st[8];
    CK(hipMemcpy(&fault, a.fault, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&mism, a.mismatch, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(st, a.stamps, 64, hipMemcpyDeviceToHost));
    printf("%-44s %s  %6.2f us/round  faults %u  mismatches %u | ticks/round: weights %5.0f slab %5.0f bar1 %5.0f reduce %5.0f bar2 %5.0f adam %5.0f bar3 %5.0f\n",
           name, uncached ? "fine-grained" : "cached      ", 1e3 * best / rounds, fault, mism, (double)st[0] / rounds, (double)st[1] / rounds,
           (double)st[2] / rounds, (double)st[3] / rounds, (double)st[4] / rounds, (double)st[5] / rounds, (double)st[6] / rounds);
    fflush(stdout);
    for (void *p : {(void *)a.slabs, (void *)a.weights, (void *)a.partials, (void *)a.flags, (void *)a.fault, (void *)a.mismatch, (void *)a.stamps}) CK(hipFree(p));
}

int main()
{
    const int R = 40;
    run<SC0 | SC1, SC0 | SC1, 1>("store sc0 sc1, load sc0 sc1", false, R);
    run<SC0 | SC1, SC0 | SC1, 16>("store sc0 sc1, load sc0 sc1, flags 64 B apart", false, R);
    run<SC0 | SC1 | NTB, SC0 | SC1 | NTB, 1>("store sc0 sc1 nt, load sc0 sc1 nt", false, R);
    run<SC1, SC1, 1>("store sc1, load sc1", false, R);
    run<NTB, SC0 | SC1, 1>("store nt, load sc0 sc1 (expect stale)", false, R);
    run<0, 0, 1>("plain store, plain load (expect stale)", false, R);
    run<0, 0, 1>("plain store, plain load", true, R);
    run<SC0 | SC1, SC0 | SC1, 1>("store sc0 sc1, load sc0 sc1", true, R);
    run<NTB, NTB, 1>("store nt, load nt", true, R);
    return 0;
}
