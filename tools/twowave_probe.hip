// Would the split-arithmetic PPO minibatch kernel (csrc/ppo_step_s3_impl.h) gain from TWO waves per SIMD on 16-sample tiles
// (v_mfma_f32_16x16x32_bf16, <= 256 registers each) instead of ONE wave per SIMD on 32-sample tiles (v_mfma_f32_32x32x16_bf16)?
// The question of round 3's review: one in-order wave serialises its GELU / operand-split vector work with its MFMAs; a second
// wave on the SIMD could fill the gaps.  This probe prices it before anybody rewrites 840 lines of kernel: the SAME work per SIMD
// in both forms --
//     BIG    4 waves x G groups of [3 ds_read_b128 (A operand) + 6 MFMA 32x32x16 + 6 P vector instructions]
//     SMALL  8 waves x G groups of [3 ds_read_b128 (A operand) + 6 MFMA 16x16x32 + 6 P / 2 vector instructions]
// (a 16x16x32 MFMA is half the flops of a 32x32x16 one, a lane of the small form owns half as many activation elements, and the
// A operand of the small form is read twice as often per flop: 16 rows x 32 k per 16 KFLOP instead of 32 rows x 16 k per 32) --
// with the vector work of kind K: 0 independent fma, 1 the GELU mix of fwd_s3 (2 transcendentals in 17), 2 the operand split.
// Also a PHASED stream: a vector-bound phase (P per MFMA) followed by a matrix-bound one (no vector work, transposing reads),
// which is the kernel's forward -> weight-gradient sequence: the two waves of a SIMD run it in lock step unless skewed.
// Prints wall cycles (max over the workgroup's waves, s_memtime) per group-pair of the SIMD, so BIG and SMALL compare directly;
// floor = 192 cycles of matrix pipe per BIG group (two SMALL groups).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/twowave_probe tools/twowave_probe.hip && tools/bin/twowave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ f32x16 mfma_big(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_small(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool SMALL> struct Acc { typedef f32x16 T; };
template <> struct Acc<true> { typedef f32x4 T; };

// PER: vector instructions behind each MFMA; KIND: 0 fma, 1 GELU mix, 2 operand split; SMALL: 16x16x32 tiles (launch 512 threads)
template <int PER, int KIND, bool SMALL>
__global__ __launch_bounds__(SMALL ? 512 : 256) void stream(long long *out, float *sink, int groups)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ROWB = 768, NT = SMALL ? 512 : 256;
    for (int e = threadIdx.x; e < 128 * ROWB / 4; e += NT) reinterpret_cast<uint32_t *>(smem)[e] = 0x3f803f80u;
    __syncthreads();
    u32x4 bh = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bm = bh, bl = bh;
    typename Acc<SMALL>::T acc0 = {0}, acc1 = {0};
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    uint32_t pk[4] = {0, 0, 0, 0};
    // A operand: BIG lane (row = lane & 31, k-half = lane >> 5); SMALL lane (row = lane & 15, k-quarter = lane >> 4); 16 bytes each,
    // chunks XOR-swizzled by the row as in csrc/split_bf16.h (swz<16>)
    const int i = SMALL ? (lane & 15) : (lane & 31), kq = SMALL ? (lane >> 4) : (lane >> 5);
    const int r0 = i & 1, r1 = (i >> 1) & 1, r2 = (i >> 2) & 1, r3 = (i >> 3) & 1;
    const int sw = ((r1 ^ r3) << 3) | (r0 << 2) | ((r1 ^ r2) << 1) | r2;
    const uint32_t base = (uint32_t)(uintptr_t)(smem + i * ROWB);
    const uint32_t x16 = 16 * (sw ^ kq);
    typedef __attribute__((address_space(3))) u32x4 *l4;
    auto issue = [&](int g, u32x4(&r)[3]) {
        const uint32_t off = SMALL ? (g & 7) * 16 * ROWB + ((((g >> 3) & 3) * 64) ^ x16) : (g & 3) * 32 * ROWB + ((((g >> 2) & 7) * 32) ^ x16);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) r[pl] = *(l4)(base + off + 256 * pl);
    };
    auto filler = [&](int slot) {
#pragma unroll
        for (int n = 0; n < PER; ++n) {
            const int q = slot * PER + n, j = q & 7;
            if (KIND == 0) {
                v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
            } else if (KIND == 1) {                    // 17-instruction GELU + GELU' mix: rcp and exp2 once each
                const int s = q % 17;
                if (s == 3) v[j] = __builtin_amdgcn_rcpf(v[j]);
                else if (s == 5) v[j] = __builtin_amdgcn_exp2f(v[j]);
                else v[j] = __builtin_fmaf(v[j], 1.0001f, v[(j + 1) & 7]);
            } else {
                switch (q % 6) {
                case 0: pk[j & 3] = pk_bf16(v[j], v[(j + 1) & 7]); break;
                case 1: v[(j + 2) & 7] = v[j] - __uint_as_float(pk[j & 3] << 16); break;
                case 2: v[(j + 3) & 7] = v[(j + 1) & 7] - __uint_as_float(pk[j & 3] & 0xffff0000u); break;
                case 3: pk[(j + 1) & 3] = pk_bf16(v[(j + 2) & 7], v[(j + 3) & 7]); break;
                case 4: v[(j + 4) & 7] = v[(j + 2) & 7] - __uint_as_float(pk[(j + 1) & 3] << 16); break;
                default: v[(j + 5) & 7] = v[(j + 3) & 7] - __uint_as_float(pk[(j + 1) & 3] & 0xffff0000u); break;
                }
            }
            asm volatile("" : "+v"(v[j]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mm = [&](u32x4 a, u32x4 b, typename Acc<SMALL>::T c) {
        if constexpr (SMALL) return mfma_small(a, b, c);
        else return mfma_big(a, b, c);
    };
    u32x4 r[2][3];
    issue(0, r[0]);
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int g = 0; g < groups; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            issue(g + u + 1, r[(u + 1) & 1]);
            const u32x4 ah = r[u][0], am = r[u][1], al = r[u][2];
            acc0 = mm(am, bm, acc0); __builtin_amdgcn_sched_barrier(0); filler(0);
            acc1 = mm(al, bh, acc1); __builtin_amdgcn_sched_barrier(0); filler(1);
            acc0 = mm(ah, bl, acc0); __builtin_amdgcn_sched_barrier(0); filler(2);
            acc1 = mm(am, bh, acc1); __builtin_amdgcn_sched_barrier(0); filler(3);
            acc0 = mm(ah, bm, acc0); __builtin_amdgcn_sched_barrier(0); filler(4);
            acc1 = mm(ah, bh, acc1); __builtin_amdgcn_sched_barrier(0); filler(5);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) { out[2 * wave] = (long long)t0; out[2 * wave + 1] = (long long)t1; }
    float s = 0.f;
    for (int r_ = 0; r_ < (SMALL ? 4 : 16); ++r_) s += acc0[r_] + acc1[r_];
    for (int n = 0; n < 8; ++n) s += v[n];
    for (int n = 0; n < 4; ++n) s += (float)pk[n];
    if (s == 12345.f) sink[0] = s;
}

// the kernel's phase sequence: `rounds` x [V groups with PER vector instructions per MFMA, then M groups with none]; SKEW: the
// second wave of every SIMD (waves 4..7 of the SMALL form) starts with the matrix-bound phase, so that the two waves of a SIMD are
// always in DIFFERENT phases (what a skewed software pipeline over two half-tiles would give)
template <int PER, bool SMALL, bool SKEW>
__global__ __launch_bounds__(SMALL ? 512 : 256) void phased(long long *out, float *sink, int rounds, int V, int M)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ROWB = 768, NT = SMALL ? 512 : 256;
    for (int e = threadIdx.x; e < 128 * ROWB / 4; e += NT) reinterpret_cast<uint32_t *>(smem)[e] = 0x3f803f80u;
    __syncthreads();
    u32x4 bh = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bm = bh, bl = bh;
    typename Acc<SMALL>::T acc0 = {0}, acc1 = {0};
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    const int i = SMALL ? (lane & 15) : (lane & 31), kq = SMALL ? (lane >> 4) : (lane >> 5);
    const int r0 = i & 1, r1 = (i >> 1) & 1, r2 = (i >> 2) & 1, r3 = (i >> 3) & 1;
    const int sw = ((r1 ^ r3) << 3) | (r0 << 2) | ((r1 ^ r2) << 1) | r2;
    const uint32_t base = (uint32_t)(uintptr_t)(smem + i * ROWB);
    const uint32_t x16 = 16 * (sw ^ kq);
    typedef __attribute__((address_space(3))) u32x4 *l4;
    auto issue = [&](int g, u32x4(&r)[3]) {
        const uint32_t off = SMALL ? (g & 7) * 16 * ROWB + ((((g >> 3) & 3) * 64) ^ x16) : (g & 3) * 32 * ROWB + ((((g >> 2) & 7) * 32) ^ x16);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) r[pl] = *(l4)(base + off + 256 * pl);
    };
    auto mm = [&](u32x4 a, u32x4 b, typename Acc<SMALL>::T c) {
        if constexpr (SMALL) return mfma_small(a, b, c);
        else return mfma_big(a, b, c);
    };
    auto group = [&](int g, u32x4(&cur)[3], u32x4(&nxt)[3], bool vec) {
        issue(g + 1, nxt);
        const u32x4 ah = cur[0], am = cur[1], al = cur[2];
        auto filler = [&](int slot) {
            if (vec) {
#pragma unroll
                for (int n = 0; n < PER; ++n) {
                    const int q = slot * PER + n, j = q & 7, s = q % 17;
                    if (s == 3) v[j] = __builtin_amdgcn_rcpf(v[j]);
                    else if (s == 5) v[j] = __builtin_amdgcn_exp2f(v[j]);
                    else v[j] = __builtin_fmaf(v[j], 1.0001f, v[(j + 1) & 7]);
                    asm volatile("" : "+v"(v[j]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        acc0 = mm(am, bm, acc0); __builtin_amdgcn_sched_barrier(0); filler(0);
        acc1 = mm(al, bh, acc1); __builtin_amdgcn_sched_barrier(0); filler(1);
        acc0 = mm(ah, bl, acc0); __builtin_amdgcn_sched_barrier(0); filler(2);
        acc1 = mm(am, bh, acc1); __builtin_amdgcn_sched_barrier(0); filler(3);
        acc0 = mm(ah, bm, acc0); __builtin_amdgcn_sched_barrier(0); filler(4);
        acc1 = mm(ah, bh, acc1); __builtin_amdgcn_sched_barrier(0); filler(5);
    };
    u32x4 r[2][3];
    issue(0, r[0]);
    unsigned long long t0, t1;
    const bool second = SKEW && wave >= 4;                  // wave-uniform
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int rd = 0; rd < rounds; ++rd) {
        if (!second) {
            for (int g = 0; g < V; g += 2) { group(g, r[0], r[1], true); group(g + 1, r[1], r[0], true); }
            for (int g = 0; g < M; g += 2) { group(g, r[0], r[1], false); group(g + 1, r[1], r[0], false); }
        } else {
            for (int g = 0; g < M; g += 2) { group(g, r[0], r[1], false); group(g + 1, r[1], r[0], false); }
            for (int g = 0; g < V; g += 2) { group(g, r[0], r[1], true); group(g + 1, r[1], r[0], true); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) { out[2 * wave] = (long long)t0; out[2 * wave + 1] = (long long)t1; }
    float s = 0.f;
    for (int r_ = 0; r_ < (SMALL ? 4 : 16); ++r_) s += acc0[r_] + acc1[r_];
    for (int n = 0; n < 8; ++n) s += v[n];
    if (s == 12345.f) sink[0] = s;
}

static double span(const long long *h, int waves)
{
    long long lo = h[0], hi = h[1];
    for (int w = 1; w < waves; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
    return (double)(hi - lo);
}

template <int PER, int KIND>
void run_stream(const char *name)
{
    long long *d; float *s;
    CK(hipMalloc(&d, 256)); CK(hipMalloc(&s, 64));
    const int groups = 1024;
    const size_t lds = 128 * 768;
    long long h[16];
    CK(hipFuncSetAttribute((const void *)stream<PER, KIND, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)stream<PER / 2, KIND, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((stream<PER, KIND, false>), dim3(1), dim3(256), lds, 0, d, s, groups);
    CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
    const double big = span(h, 4) / groups;
    // SMALL: every wave runs the same number of groups (16 samples: two 16-feature tiles x one 32-deep k-step per BIG tile pair of k-steps),
    // each half the flops; two waves per SIMD make up the BIG wave's work
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((stream<PER / 2, KIND, true>), dim3(1), dim3(512), lds, 0, d, s, groups);
    CK(hipMemcpy(h, d, 128, hipMemcpyDeviceToHost));
    const double small = span(h, 8) / groups;
    printf("F  %-22s %2d per 32x32x16 MFMA: one wave/SIMD %6.1f   two waves/SIMD on 16x16x32 %6.1f cycles per 6-MFMA-group equivalent  (%+.1f %%)\n",
           name, PER, big, small, 100.0 * (small - big) / big);
    CK(hipFree(d)); CK(hipFree(s));
}

template <int PER>
void run_phased()
{
    long long *d; float *s;
    CK(hipMalloc(&d, 256)); CK(hipMalloc(&s, 64));
    const int rounds = 16, V = 32, M = 32;
    const size_t lds = 128 * 768;
    long long h[16];
    CK(hipFuncSetAttribute((const void *)phased<PER, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)phased<PER / 2, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)phased<PER / 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((phased<PER, false, false>), dim3(1), dim3(256), lds, 0, d, s, rounds, V, M);
    CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
    const double n = (double)rounds * (V + M);
    const double big = span(h, 4) / n;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((phased<PER / 2, true, false>), dim3(1), dim3(512), lds, 0, d, s, rounds, V, M);
    CK(hipMemcpy(h, d, 128, hipMemcpyDeviceToHost));
    const double lock = span(h, 8) / n;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((phased<PER / 2, true, true>), dim3(1), dim3(512), lds, 0, d, s, rounds, V, M);
    CK(hipMemcpy(h, d, 128, hipMemcpyDeviceToHost));
    const double skew = span(h, 8) / n;
    printf("G  vector phase (%2d per MFMA) + matrix phase, equal lengths: one wave/SIMD %6.1f   two waves in lock step %6.1f (%+.1f %%)   two waves, phases "
           "opposed %6.1f (%+.1f %%)   [cycles per 6-MFMA-group equivalent, floor 192]\n", PER, big, lock, 100.0 * (lock - big) / big, skew,
           100.0 * (skew - big) / big);
    CK(hipFree(d)); CK(hipFree(s));
}

int main()
{
    run_stream<0, 0>("no vector work");
    run_stream<4, 0>("independent fma");
    run_stream<6, 0>("independent fma");
    run_stream<8, 0>("independent fma");
    run_stream<12, 0>("independent fma");
    run_stream<4, 1>("GELU mix");
    run_stream<6, 1>("GELU mix");
    run_stream<8, 1>("GELU mix");
    run_stream<12, 1>("GELU mix");
    run_stream<6, 2>("operand split");
    run_stream<12, 2>("operand split");
    run_phased<6>();
    run_phased<8>();
    run_phased<12>();
    return 0;
}
