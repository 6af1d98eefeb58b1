// What clock does THIS box give a kernel?  (round 5: the pool's boxes run the PPO minibatch kernel at different speeds; the chip
// clocks to its power budget -- MI355X_MICROARCH.md, "DVFS give-back" -- so the question is asked of the hardware, per kind of work.)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/clock_probe tools/clock_probe.hip -lhsa-runtime64 && tools/bin/clock_probe
//
// Every workgroup (one per CU, 4 waves = one per SIMD, like the minibatch kernel) runs a fixed instruction count of ONE kind of work
// and stamps s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around it; the ratio is the clock the body ran at, the
// cycles per instruction say whether the pipe itself is as fast as on another box.  Bodies: bf16 MFMA 32x32x16 back to back on two
// accumulators (the minibatch kernel's matrix instruction), fp32 MFMA 32x32x2, plain fp32 FMA, an MFMA + 5 VALU mix (the minibatch
// kernel's forward layers), and LDS reads.  Output: one JSON line.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ unsigned long long memtime()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

struct Rec {
    unsigned long long cyc, wall;
};

template <int KIND>
__global__ __launch_bounds__(256) void body(int iters, Rec *out, float *sink)
{
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0};
    bf16x8 p = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
    bf16x8 q = {0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00, 0x3c00};
    float v0 = lane * 1e-3f, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f, v4 = 0.125f;
    const unsigned long long w0 = wall_clock64(), c0 = memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {                 // bf16 MFMA 32x32x16, two accumulators, 16 per iteration
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q, p, a1, 0, 0, 0);
            }
        } else if (KIND == 1) {          // fp32 MFMA 32x32x2, 16 per iteration
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, v2, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, v1, a1, 0, 0, 0);
            }
        } else if (KIND == 2) {          // fp32 FMA, 5 independent chains, 80 per iteration
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v0 = __builtin_fmaf(v0, 0.999f, 1e-3f);
                v1 = __builtin_fmaf(v1, 0.998f, 2e-3f);
                v2 = __builtin_fmaf(v2, 0.997f, 3e-3f);
                v3 = __builtin_fmaf(v3, 0.996f, 4e-3f);
                v4 = __builtin_fmaf(v4, 0.995f, 5e-3f);
            }
        } else if (KIND == 3) {          // the forward layers' mix: one bf16 MFMA + 5 VALU, 16 groups per iteration
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                v0 = __builtin_fmaf(v0, 0.999f, 1e-3f);
                v1 = __builtin_fmaf(v1, 0.998f, 2e-3f);
                v2 = __builtin_fmaf(v2, 0.997f, 3e-3f);
                v3 = __builtin_fmaf(v3, 0.996f, 4e-3f);
                v4 = __builtin_fmaf(v4, 0.995f, 5e-3f);
                __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q, p, a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                v0 = __builtin_fmaf(v0, 0.999f, 1e-3f);
                v1 = __builtin_fmaf(v1, 0.998f, 2e-3f);
                v2 = __builtin_fmaf(v2, 0.997f, 3e-3f);
                v3 = __builtin_fmaf(v3, 0.996f, 4e-3f);
                v4 = __builtin_fmaf(v4, 0.995f, 5e-3f);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {                         // LDS: 16 ds_read_b128 per iteration (dependent on the loop counter only)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float4 x = *reinterpret_cast<const float4 *>(&lds[((lane * 4 + 64 * u + 4 * it) & 4092)]);
                v0 += x.x; v1 += x.y; v2 += x.z; v3 += x.w;
            }
        }
    }
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4));
    const unsigned long long c1 = memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{c1 - c0, w1 - w0};
    float s = v0 + v1 + v2 + v3 + v4;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    if (s == 12345.678f) sink[0] = s;
}

// KIND 5: the minibatch kernel's forward-layer shape -- per group 3 ds_read_b128 sweeping 152 KB of dynamic LDS (its weight images), 6 bf16
// MFMAs on two accumulators, 30 vector instructions with transcendentals among them (its GELU stages), one wave per SIMD (the 152 KB
// keep a second workgroup off the CU).  If a box runs THIS slower than its neighbours while the single-instruction bodies above agree,
// the difference is in how the pieces share the CU, not in any one pipe.
__global__ __launch_bounds__(256) void body_k6like(int iters, Rec *out, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) float big[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 38912; i += 256) big[i] = (float)i * 1e-6f;
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0};
    float v[6] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f};
    const unsigned long long w0 = wall_clock64(), c0 = memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int base = ((it * 4 + u) * 3 * 256 + lane * 4) % 38000;
            const float4 x0 = *reinterpret_cast<const float4 *>(&big[(base) & ~3]);
            const float4 x1 = *reinterpret_cast<const float4 *>(&big[(base + 256) & ~3]);
            const float4 x2 = *reinterpret_cast<const float4 *>(&big[(base + 512) & ~3]);
            bf16x8 p = __builtin_bit_cast(bf16x8, x0), q = __builtin_bit_cast(bf16x8, x1), r = __builtin_bit_cast(bf16x8, x2);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m == 0 ? p : (m == 1 ? q : r), q, a0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(v[k], 0.999f, 1e-3f);
                v[4] = __builtin_amdgcn_rcpf(v[4] + 1.5f);
                __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, m == 0 ? r : (m == 1 ? p : q), a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(v[k], 0.998f, 2e-3f);
                v[5] = __builtin_amdgcn_exp2f(-(v[5] * v[5]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("" : "+v"(a0), "+v"(a1));
    const unsigned long long c1 = memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{c1 - c0, w1 - w0};
    float s = v[0] + v[1] + v[2] + v[3] + v[4] + v[5];
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    if (s == 12345.678f) sink[0] = s;
}

void run_k6like(int khz, int iters, int reps)
{
    Rec *d;
    float *sink;
    const int grid = 256;
    const size_t lds = 38912 * sizeof(float);                  // 152 KB
    CHECK(hipFuncSetAttribute((const void *)body_k6like, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipMalloc(&d, grid * sizeof(Rec)));
    CHECK(hipMalloc(&sink, 4));
    std::vector<Rec> h(grid);
    double sum_mhz = 0, sum_cyc = 0, mn = 1e30, mx = 0;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(body_k6like, dim3(grid), dim3(256), lds, 0, iters, d, sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost));
        double cyc = 0, wall = 0;
        for (auto &x : h) { cyc += (double)x.cyc; wall += (double)x.wall; if (r >= reps / 2) { mn = std::min(mn, (double)x.cyc); mx = std::max(mx, (double)x.cyc); } }
        if (r >= reps / 2) { sum_mhz += cyc / wall * khz * 1e-3; sum_cyc += cyc / grid; }
    }
    const int n = reps - reps / 2;
    printf("\"k6_like_forward_mix\": {\"shader_mhz\": %.1f, \"cycles_per_group\": %.1f, \"workgroup_cycles_min_over_mean\": %.3f, "
           "\"workgroup_cycles_max_over_mean\": %.3f, \"us_per_launch\": %.1f}, ",
           sum_mhz / n, sum_cyc / n / (iters * 4.0), mn / (sum_cyc / n), mx / (sum_cyc / n), sum_cyc / n / (sum_mhz / n));
    CHECK(hipFree(d));
    CHECK(hipFree(sink));
}

// instruction fetch: a workgroup's waves walk 56 KB of straight-line VALU code (one dependent fp32 op per 4 bytes, ~4 cycles each: 56 KB
// in ~27 us -- the minibatch kernel's rate: ~55 KB per network in ~33 us), one workgroup per CU (155 KB of LDS asked for), 256 workgroups.
//   single     every workgroup walks block A
//   dual_map0  workgroups with blockIdx.y == 0 walk block A, the others block B: both blocks on every XCD (the minibatch kernel's old map)
//   dual_map1  linear id % 8 < 4 walks A, the rest B: one block per XCD (its map 1)
//   dual_map2  linear id % 32 < 16 walks A, the rest B: one block per pair of shader engines, both blocks on every XCD (its map 2)
//   single_112KB  every workgroup walks A, then B (twice the work: compare with 2 x single)
// Instruction caches are 64 KB and shared by neighbouring CUs: a box where dual_map0 is slower than single / dual_map1 is a box that pays
// for two code paths per instruction cache -- with no MFMA, no LDS traffic and no memory traffic in the picture.
#define CODE_BLOCK(ins) asm volatile(".rept 14336\n" ins " %0, %0, %0\n.endr" : "+v"(x))
template <int MODE>
__global__ __launch_bounds__(256) void code_walk(float *sink, unsigned long long *rec)
{
    extern __shared__ float lds_pad[];
    float x = (float)threadIdx.x;
    const int L = (int)(blockIdx.x + gridDim.x * blockIdx.y);
    const bool a = MODE == 0 ? true : MODE == 1 ? blockIdx.y == 0 : MODE == 4 ? (L & 31) < 16 : (L & 7) < 4;
    const unsigned long long w0 = wall_clock64();
    if (MODE == 3) { CODE_BLOCK("v_add_f32"); CODE_BLOCK("v_mul_f32"); }      // every workgroup walks both: 112 KB through each instruction cache
    else if (a) CODE_BLOCK("v_add_f32");
    else CODE_BLOCK("v_mul_f32");
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) { rec[2 * L] = w0; rec[2 * L + 1] = w1; }
    if (x == 12345.678f) { lds_pad[threadIdx.x] = x; sink[0] = lds_pad[0]; }
}

// the instruction caches' capacity: ONE block of KB kilobytes walked by every workgroup, back to back -- microseconds per KB stay flat while
// the block fits and rise where it does not (visibly so only on a box whose instruction-cache miss path is slow)
#define STR2(x) #x
#define STR(x) STR2(x)
template <int KB>
__global__ __launch_bounds__(256) void code_size_walk(float *sink)
{
    extern __shared__ float lds_pad[];
    float x = (float)threadIdx.x;
    static_assert(KB == 40 || KB == 48 || KB == 56 || KB == 60 || KB == 64 || KB == 68 || KB == 72 || KB == 80 || KB == 96, "sizes");
    if (KB == 40) asm volatile(".rept 10240\n v_add_f32 %0, %0, %0\n.endr" : "+v"(x));
    if (KB == 48) asm volatile(".rept 12288\n v_add_f32 %0, %0, %0\n.endr" : "+v"(x));
    if (KB == 56) asm volatile(".rept 14336\n v_add_f32 %0, %0, %0\n.endr" : "+v"(x));
    if (KB == 60) asm volatile(".rept 15360\n v_add_f32 %0, %0, %0\n.endr" : "+v"(x));
    if (KB == 64) asm volatile(".rept 16384\n v_add_f32 %0, %0, %0\n.endr" : "+v"(x));
    if (KB == 68) asm volatile(".rept 17408\n v_add_f32 %0, %0, %0\n.endr" : "+v"(x));
    if (KB == 72) asm volatile(".rept 18432\n v_add_f32 %0, %0, %0\n.endr" : "+v"(x));
    if (KB == 80) asm volatile(".rept 20480\n v_add_f32 %0, %0, %0\n.endr" : "+v"(x));
    if (KB == 96) asm volatile(".rept 24576\n v_add_f32 %0, %0, %0\n.endr" : "+v"(x));
    if (x == 12345.678f) { lds_pad[threadIdx.x] = x; sink[0] = lds_pad[0]; }
}

template <int KB>
double run_code_size(float *sink)
{
    const int lds = 155 * 1024;
    CHECK(hipFuncSetAttribute((const void *)code_size_walk<KB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(code_size_walk<KB>, dim3(128, 2), dim3(256), lds, 0, sink);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(code_size_walk<KB>, dim3(128, 2), dim3(256), lds, 0, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / 20;
}

void run_code_sizes()
{
    float *sink;
    CHECK(hipMalloc(&sink, 4));
    printf("\"code_walk_us_by_KB\": {\"40\": %.1f, ", run_code_size<40>(sink));
    printf("\"48\": %.1f, ", run_code_size<48>(sink));
    printf("\"56\": %.1f, ", run_code_size<56>(sink));
    printf("\"60\": %.1f, ", run_code_size<60>(sink));
    printf("\"64\": %.1f, ", run_code_size<64>(sink));
    printf("\"68\": %.1f, ", run_code_size<68>(sink));
    printf("\"72\": %.1f, ", run_code_size<72>(sink));
    printf("\"80\": %.1f, ", run_code_size<80>(sink));
    printf("\"96\": %.1f}, ", run_code_size<96>(sink));
    CHECK(hipFree(sink));
}

template <int MODE>
void run_code_walk(const char *name, int khz, bool last)
{
    const int lds = 155 * 1024, grid = 128;
    CHECK(hipFuncSetAttribute((const void *)code_walk<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    float *sink;
    unsigned long long *rec;
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&rec, 2 * 256 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(512);
    auto one = [&](double &span_us, double &wg_us) {
        hipLaunchKernelGGL(code_walk<MODE>, dim3(grid, 2), dim3(256), lds, 0, sink, rec);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long lo = ~0ull, hi = 0;
        double sum = 0;
        for (int w = 0; w < 256; ++w) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); sum += (double)(h[2 * w + 1] - h[2 * w]); }
        span_us = (double)(hi - lo) / khz * 1e3;
        wg_us = sum / 256 / khz * 1e3;
    };
    double first_span, first_wg, span = 0, wg = 0;
    one(first_span, first_wg);
    for (int r = 0; r < 3; ++r) { double s_, w_; one(s_, w_); }
    // back to back on the stream (no host round trip between launches), HIP events around 20
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(code_walk<MODE>, dim3(grid, 2), dim3(256), lds, 0, sink, rec);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    for (int r = 0; r < 5; ++r) { double s_, w_; one(s_, w_); span += s_ / 5; wg += w_ / 5; }
    printf("\"%s\": {\"first_launch_span_us\": %.1f, \"span_us\": %.1f, \"workgroup_us\": %.1f, \"back_to_back_us\": %.1f}%s", name, first_span, span, wg,
           ms * 1e3 / 20, last ? "}, " : ", ");
    CHECK(hipFree(sink));
    CHECK(hipFree(rec));
}

// where the loader put this process's kernel code: a kernel reports its program counter, ROCr says which pool owns that address (a GPU's
// local memory or the host's), how large the allocation is and whether it is fine-grained (uncached in the GPU's L2)
__global__ void where_am_i(unsigned long long *out)
{
    unsigned long long pc;
    asm volatile("s_getpc_b64 %0" : "=s"(pc));
    out[0] = pc;
}

void run_code_memory()
{
    unsigned long long *o, pc = 0;
    CHECK(hipMalloc(&o, 8));
    hipLaunchKernelGGL(where_am_i, dim3(1), dim3(1), 0, 0, o);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&pc, o, 8, hipMemcpyDeviceToHost));
    CHECK(hipFree(o));
    hsa_amd_pointer_info_t info;
    memset(&info, 0, sizeof(info));
    info.size = sizeof(info);
    (void)hsa_init();
    const hsa_status_t st = hsa_amd_pointer_info((const void *)pc, &info, nullptr, nullptr, nullptr);
    hsa_device_type_t dt = (hsa_device_type_t)-1;
    if (st == HSA_STATUS_SUCCESS && info.type != HSA_EXT_POINTER_TYPE_UNKNOWN) (void)hsa_agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &dt);
    printf("\"code_memory\": {\"pc\": \"0x%llx\", \"hsa_status\": %d, \"pointer_type\": %d, \"allocation_bytes\": %zu, \"owner\": \"%s\", \"global_flags\": %u, "
           "\"note\": \"global_flags: 1 kernarg, 2 fine-grained, 4 coarse-grained, 8 extended-scope fine-grained\"}, ",
           pc, (int)st, (int)info.type, info.sizeInBytes, dt == HSA_DEVICE_TYPE_GPU ? "gpu" : dt == HSA_DEVICE_TYPE_CPU ? "cpu" : "unknown", info.global_flags);
}

void run_code_walks(int khz)
{
    printf("\"code_walk_56KB\": {");
    run_code_walk<0>("single", khz, false);
    run_code_walk<1>("dual_map0", khz, false);
    run_code_walk<2>("dual_map1", khz, false);
    run_code_walk<4>("dual_map2", khz, false);
    run_code_walk<3>("single_112KB", khz, true);
}

template <int KIND>
void run(const char *name, int per_iter, int iters, int reps, int khz, bool last)
{
    Rec *d;
    float *sink;
    const int grid = 256;
    CHECK(hipMalloc(&d, grid * sizeof(Rec)));
    CHECK(hipMalloc(&sink, 4));
    std::vector<Rec> h(grid);
    double best_mhz = 0, sum_mhz = 0, sum_cpi = 0, first_mhz = 0;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(body<KIND>, dim3(grid), dim3(256), 0, 0, iters, d, sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost));
        double cyc = 0, wall = 0;
        for (auto &x : h) { cyc += (double)x.cyc; wall += (double)x.wall; }
        const double mhz = cyc / wall * khz * 1e-3, cpi = cyc / grid / ((double)iters * per_iter);
        if (r == 0) first_mhz = mhz;
        if (r >= reps / 2) { sum_mhz += mhz; sum_cpi += cpi; }
        best_mhz = mhz > best_mhz ? mhz : best_mhz;
    }
    const int n = reps - reps / 2;
    printf("\"%s\": {\"shader_mhz\": %.1f, \"shader_mhz_first_launch\": %.1f, \"shader_mhz_max\": %.1f, \"cycles_per_instruction\": %.2f, "
           "\"us_per_launch\": %.1f}%s",
           name, sum_mhz / n, first_mhz, best_mhz, sum_cpi / n, sum_cpi / n * iters * per_iter / (sum_mhz / n), last ? "" : ", ");
    CHECK(hipFree(d));
    CHECK(hipFree(sink));
}

int main()
{
    int dev = 0, khz = 0, cus = 0, sclk = 0, mclk = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CHECK(hipDeviceGetAttribute(&sclk, hipDeviceAttributeClockRate, dev));
    CHECK(hipDeviceGetAttribute(&mclk, hipDeviceAttributeMemoryClockRate, dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"wall_clock_khz\": %d, \"attr_clock_khz\": %d, \"attr_memory_clock_khz\": %d, ",
           prop.name, prop.gcnArchName, cus, khz, sclk, mclk);
    // ~100-200 us per launch, 24 launches each: long enough for the power manager to settle on the body's clock
    run_code_memory();
    run_code_walks(khz);
    run_code_sizes();
    run_k6like(khz, 120, 24);
    run<0>("mfma_bf16_32x32x16", 16, 400, 24, khz, false);
    run<1>("mfma_f32_32x32x2", 16, 200, 24, khz, false);
    run<2>("valu_fma_f32", 80, 600, 24, khz, false);
    run<3>("mfma_bf16_plus_5_valu", 16, 300, 24, khz, false);
    run<4>("lds_read_b128", 16, 1000, 24, khz, true);
    printf("}\n");
    return 0;
}
