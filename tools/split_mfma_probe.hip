// Ground facts for an fp32-equivalent MLP step on the bf16 matrix pipe (operands split into three bf16 parts, fp32 accumulate):
//   A  operand / result layout of v_mfma_f32_32x32x16_bf16 (checked against a host product),
//   B  what ds_read_b64_tr_b16 returns for per-lane addresses (lane -> LDS element map printed),
//   C  accuracy of the 6-term and 9-term split products against fp64, next to the fp32 MFMA (v_mfma_f32_32x32x2_f32),
//   D  cycles per (6 MFMA + operand reads + hidden VALU) group for one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/split_mfma_probe tools/split_mfma_probe.hip && tools/bin/split_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b)      // v_cvt_pk_bf16_f32 (RNE): a in the low half
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float lo_f32(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f32(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// x = h + m + l, each part a bf16 (RNE); two elements at a time, packed
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = pk_bf16(x0, x1);
    const float r0 = x0 - lo_f32(h), r1 = x1 - hi_f32(h);
    m = pk_bf16(r0, r1);
    const float q0 = r0 - lo_f32(m), q1 = r1 - hi_f32(m);
    l = pk_bf16(q0, q1);
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- A + C: C (32 x 32) = A (32 x K) . B (K x 32), row-major fp32 inputs, K % 16 == 0.  mode 0: fp32 MFMA, 6: split 6 terms, 9: 9 terms
template <int MODE>
__global__ void gemm32(const float *A, const float *B, float *C, int K)
{
    const int lane = threadIdx.x, i = lane & 31, kb = lane >> 5;
    f32x16 acc = {0};
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + kb], B[(k + kb) * 32 + i], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            u32x4 ah, am, al, bh, bm, bl;
            for (int e = 0; e < 4; ++e) {
                uint32_t h, m, l;
                split2(A[i * K + k0 + 8 * kb + 2 * e], A[i * K + k0 + 8 * kb + 2 * e + 1], h, m, l);
                ah[e] = h; am[e] = m; al[e] = l;
                split2(B[(k0 + 8 * kb + 2 * e) * 32 + i], B[(k0 + 8 * kb + 2 * e + 1) * 32 + i], h, m, l);
                bh[e] = h; bm[e] = m; bl[e] = l;
            }
            if (MODE == 9) {
                acc = mfma_bf16(al, bl, acc);
                acc = mfma_bf16(al, bm, acc);
                acc = mfma_bf16(am, bl, acc);
            }
            acc = mfma_bf16(am, bm, acc);
            acc = mfma_bf16(al, bh, acc);
            acc = mfma_bf16(ah, bl, acc);
            acc = mfma_bf16(am, bh, acc);
            acc = mfma_bf16(ah, bm, acc);
            acc = mfma_bf16(ah, bh, acc);
        }
    }
    for (int r = 0; r < 16; ++r) C[(8 * (r >> 2) + 4 * kb + (r & 3)) * 32 + i] = acc[r];
}

// ---- B: transpose read.  LDS holds ushort e at element e.  mode 0: every lane the same address; 1: lane l -> byte 8 l;
//      2: lane l -> row (l & 15) >> 2 ... the [4][16] block recipe: byte ((l >> 4) * 4 + ((l & 15) >> 2)) * ROWB + 8 * (l & 3)
__global__ void tr_probe(int mode, int rowb, int *out)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int e = threadIdx.x; e < 8192; e += 64) lds[e] = (unsigned short)e;
    __syncthreads();
    const int l = threadIdx.x;
    int byte = 0;
    if (mode == 1) byte = 8 * l;
    if (mode == 2) byte = ((l >> 4) * 4 + ((l & 15) >> 2)) * rowb + 8 * (l & 3);
    typedef __attribute__((address_space(3))) s16x4 *lptr;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)((__attribute__((address_space(3))) char *)lds + byte));
    for (int j = 0; j < 4; ++j) out[4 * l + j] = (unsigned short)v[j];
}

// ---- D: timing.  One wave per SIMD (512 registers requested through the launch bounds), LDS images of three planes;
//      per group: 6 operand reads (b64 or transposing) issued ONE GROUP AHEAD, NV extra VALU instructions, 6 MFMAs into NACC
//      accumulators in turn
template <int NV, bool TR, int NACC>
__global__ __launch_bounds__(256) void mfma_time(long long *out, float *sink, int groups)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, hi = lane >> 5;
    constexpr int ROWB = 776;
    for (int e = threadIdx.x; e < 128 * ROWB / 4; e += 256) reinterpret_cast<uint32_t *>(smem)[e] = 0x3f803f80u;
    __syncthreads();
    u32x4 bh = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bm = bh, bl = bh;
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x16{0};
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    typedef __attribute__((address_space(3))) s16x4 *lptr;
    const uint32_t base = (uint32_t)(uintptr_t)(smem + i * ROWB + 8 * hi);
    const uint32_t qbase = (uint32_t)(uintptr_t)(smem + ((lane >> 5) * 8 + ((lane & 15) >> 2)) * ROWB + 32 * ((lane >> 4) & 1) + 8 * (lane & 3));
    auto issue = [&](int g, uint2(&r)[6]) {
        const uint32_t off = (g & 3) * 32 * ROWB + ((g >> 2) & 7) * 32;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            if (TR) {
                r[2 * pl] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(qbase + off + 256 * pl)));
                r[2 * pl + 1] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(qbase + off + 256 * pl + 4 * ROWB)));
            } else {
                typedef __attribute__((address_space(3))) u32x2 *l2;
                r[2 * pl] = __builtin_bit_cast(uint2, *(l2)(base + off + 256 * pl));
                r[2 * pl + 1] = __builtin_bit_cast(uint2, *(l2)(base + off + 256 * pl + 16));
            }
        }
    };
    uint2 r[2][6];
    issue(0, r[0]);
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int g = 0; g < groups; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            issue(g + u + 1, r[(u + 1) & 1]);
            const uint2(&c)[6] = r[u];
            const u32x4 ah = {c[0].x, c[0].y, c[1].x, c[1].y}, am = {c[2].x, c[2].y, c[3].x, c[3].y}, al = {c[4].x, c[4].y, c[5].x, c[5].y};
            acc[0] = mfma_bf16(am, bm, acc[0]);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                v[n & 7] = __builtin_fmaf(v[n & 7], 1.0001f, 0.5f);
                asm volatile("" : "+v"(v[n & 7]));
            }
            acc[1 % NACC] = mfma_bf16(al, bh, acc[1 % NACC]);
            acc[2 % NACC] = mfma_bf16(ah, bl, acc[2 % NACC]);
            acc[3 % NACC] = mfma_bf16(am, bh, acc[3 % NACC]);
            acc[4 % NACC] = mfma_bf16(ah, bm, acc[4 % NACC]);
            acc[5 % NACC] = mfma_bf16(ah, bh, acc[5 % NACC]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[wave] = (long long)(t1 - t0);
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r_ = 0; r_ < 16; ++r_) s += acc[n][r_];
    for (int n = 0; n < 8; ++n) s += v[n];
    if (s == 12345.f) sink[0] = s;
}

// ---- E: the same stream with the vector work DISTRIBUTED: PER instructions behind each of the 6 MFMAs of a group (two
//      accumulators in turn), of kind KIND: 0 independent v_fma_f32 chains, 1 the operand split (cvt_pk / shift / and / sub),
//      2 v_accvgpr_read of a finished accumulator + fma, 3 transcendentals (v_exp_f32 / v_rcp_f32) mixed 1:3 with fma,
//      4 / 5 / 6 one / two / four DEPENDENT fma chains (how much of the price is dependent-issue latency)
template <int PER, int KIND>
__global__ __launch_bounds__(256) void mfma_fill(long long *out, float *sink, int groups)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, hi = lane >> 5;
    constexpr int ROWB = 768;
    for (int e = threadIdx.x; e < 128 * ROWB / 4; e += 256) reinterpret_cast<uint32_t *>(smem)[e] = 0x3f803f80u;
    __syncthreads();
    u32x4 bh = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bm = bh, bl = bh;
    f32x16 acc0 = {0}, acc1 = {0}, old = {0};
    for (int r = 0; r < 16; ++r) old[r] = (float)(r + lane);
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    uint32_t pk[4] = {0, 0, 0, 0};
    // rows 768 bytes apart would put a 16-lane read group on ONE bank group: chunks are XOR-swizzled by the row as in
    // csrc/ppo_step_s3_impl.h (swz<16>)
    const int r0 = i & 1, r1 = (i >> 1) & 1, r2 = (i >> 2) & 1, r3 = (i >> 3) & 1;
    const int sw = ((r1 ^ r3) << 3) | (r0 << 2) | ((r1 ^ r2) << 1) | r2;
    const uint32_t base = (uint32_t)(uintptr_t)(smem + i * ROWB);
    const uint32_t x16 = 16 * (sw ^ hi);
    typedef __attribute__((address_space(3))) u32x4 *l4;
    auto issue = [&](int g, u32x4(&r)[3]) {
        const uint32_t off = (g & 3) * 32 * ROWB + ((((g >> 2) & 7) * 32) ^ x16);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) r[pl] = *(l4)(base + off + 256 * pl);
    };
    auto filler = [&](int slot) {
#pragma unroll
        for (int n = 0; n < PER; ++n) {
            const int j = (slot * PER + n) & 7;
            if (KIND == 0) {
                v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
            } else if (KIND == 1) {
                switch (n % 6) {
                case 0: pk[j & 3] = pk_bf16(v[j], v[(j + 1) & 7]); break;
                case 1: v[(j + 2) & 7] = v[j] - lo_f32(pk[j & 3]); break;
                case 2: v[(j + 3) & 7] = v[(j + 1) & 7] - hi_f32(pk[j & 3]); break;
                case 3: pk[(j + 1) & 3] = pk_bf16(v[(j + 2) & 7], v[(j + 3) & 7]); break;
                case 4: v[(j + 4) & 7] = v[(j + 2) & 7] - lo_f32(pk[(j + 1) & 3]); break;
                default: v[(j + 5) & 7] = v[(j + 3) & 7] - hi_f32(pk[(j + 1) & 3]); break;
                }
            } else if (KIND == 2) {
                v[j] = __builtin_fmaf(old[(slot * PER + n) & 15], 1.0001f, v[j]);
            } else if (KIND == 3) {
                if ((n & 3) == 0) v[j] = __builtin_amdgcn_exp2f(v[j]);
                else v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
            } else {      // KIND 4 / 5 / 6: 1 / 2 / 4 dependent chains
                const int c = (slot * PER + n) % (KIND == 4 ? 1 : KIND == 5 ? 2 : 4);
                v[c] = __builtin_fmaf(v[c], 1.0001f, 0.5f);
                asm volatile("" : "+v"(v[c]));
            }
            asm volatile("" : "+v"(v[j]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    u32x4 r[2][3];
    issue(0, r[0]);
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int g = 0; g < groups; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            issue(g + u + 1, r[(u + 1) & 1]);
            const u32x4 ah = r[u][0], am = r[u][1], al = r[u][2];
            acc0 = mfma_bf16(am, bm, acc0); __builtin_amdgcn_sched_barrier(0); filler(0);
            acc1 = mfma_bf16(al, bh, acc1); __builtin_amdgcn_sched_barrier(0); filler(1);
            acc0 = mfma_bf16(ah, bl, acc0); __builtin_amdgcn_sched_barrier(0); filler(2);
            acc1 = mfma_bf16(am, bh, acc1); __builtin_amdgcn_sched_barrier(0); filler(3);
            acc0 = mfma_bf16(ah, bm, acc0); __builtin_amdgcn_sched_barrier(0); filler(4);
            acc1 = mfma_bf16(ah, bh, acc1); __builtin_amdgcn_sched_barrier(0); filler(5);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[wave] = (long long)(t1 - t0);
    float s = 0.f;
    for (int r_ = 0; r_ < 16; ++r_) s += acc0[r_] + acc1[r_];
    for (int n = 0; n < 8; ++n) s += v[n];
    for (int n = 0; n < 4; ++n) s += (float)pk[n];
    if (s == 12345.f) sink[0] = s;
}

template <int PER, int KIND>
void time_fill(const char *name)
{
    long long *d; float *s;
    CK(hipMalloc(&d, 64)); CK(hipMalloc(&s, 64));
    const int groups = 1024;
    const size_t lds = 128 * 768;
    CK(hipFuncSetAttribute((const void *)mfma_fill<PER, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((mfma_fill<PER, KIND>), dim3(1), dim3(256), lds, 0, d, s, groups);
    long long h[4];
    CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    printf("E  %-34s %d per MFMA: %6.1f cycles per group of 6 MFMAs  (%+.1f per filler instruction over 206)\n", name, PER, (double)h[0] / groups,
           PER ? ((double)h[0] / groups - 206.0) / (6 * PER) : 0.0);
    CK(hipFree(d)); CK(hipFree(s));
}

template <int NV, bool TR, int NACC>
void time_it(const char *name)
{
    long long *d; float *s;
    CK(hipMalloc(&d, 64)); CK(hipMalloc(&s, 64));
    const int groups = 1024;
    const size_t lds = 128 * 776;
    CK(hipFuncSetAttribute((const void *)mfma_time<NV, TR, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((mfma_time<NV, TR, NACC>), dim3(1), dim3(256), lds, 0, d, s, groups);
    long long h[4];
    CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    printf("D  %-40s %6.1f cycles per group of 6 MFMAs (wave 0), %6.1f (wave 3)   [6 x 32 = 192 is the pipe's floor]\n", name, (double)h[0] / groups,
           (double)h[3] / groups);
    CK(hipFree(d)); CK(hipFree(s));
}

int main()
{
    // ---- A + C
    for (int K : {16, 64, 128}) {
        std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
        srand(1234 + K);
        auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0) * (1.f + (rand() % 7)); };
        for (auto &x : A) x = rnd();
        for (auto &x : B) x = rnd();
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        for (int mode : {0, 6, 9}) {
            if (mode == 0) hipLaunchKernelGGL(gemm32<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            if (mode == 6) hipLaunchKernelGGL(gemm32<6>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            if (mode == 9) hipLaunchKernelGGL(gemm32<9>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0, mean = 0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; mag += fabs((double)A[i * K + k] * B[k * 32 + j]); }
                    const double e = fabs(C[i * 32 + j] - ref) / mag;
                    worst = fmax(worst, e); mean += e / 1024;
                }
            printf("A/C K=%3d  %-28s error / sum|a b|: max %.3e  mean %.3e  (2^-24 = 5.96e-08)\n", K,
                   mode == 0 ? "fp32 MFMA 32x32x2" : mode == 6 ? "bf16 split, 6 terms" : "bf16 split, 9 terms", worst, mean);
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    // ---- B
    int *dout;
    CK(hipMalloc(&dout, 256 * 4));
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, mode, 64, dout);
        int h[256];
        CK(hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost));
        printf("B  mode %d (0: uniform address, 1: lane l -> byte 8 l, 2: [4][16] block recipe with 64-byte rows): lane: elements\n", mode);
        for (int l = 0; l < 64; ++l) printf("   %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
    }
    CK(hipFree(dout));
    // ---- D
    time_it<0, false, 1>("b64 reads, no VALU, 1 acc");
    time_it<0, false, 2>("b64 reads, no VALU, 2 acc");
    time_it<0, false, 3>("b64 reads, no VALU, 3 acc");
    time_it<12, false, 1>("b64 reads, 12 VALU, 1 acc");
    time_it<24, false, 1>("b64 reads, 24 VALU, 1 acc");
    time_it<24, false, 2>("b64 reads, 24 VALU, 2 acc");
    time_it<36, false, 2>("b64 reads, 36 VALU, 2 acc");
    time_it<0, true, 1>("tr reads, no VALU, 1 acc");
    time_it<0, true, 2>("tr reads, no VALU, 2 acc");
    time_it<24, true, 2>("tr reads, 24 VALU, 2 acc");
    time_fill<0, 0>("no filler");
    time_fill<2, 0>("independent fma");
    time_fill<4, 0>("independent fma");
    time_fill<6, 0>("independent fma");
    time_fill<8, 0>("independent fma");
    time_fill<12, 0>("independent fma");
    time_fill<4, 1>("operand split");
    time_fill<6, 1>("operand split");
    time_fill<8, 1>("operand split");
    time_fill<4, 2>("accumulator read + fma");
    time_fill<6, 2>("accumulator read + fma");
    time_fill<4, 4>("one dependent fma chain");
    time_fill<6, 4>("one dependent fma chain");
    time_fill<4, 5>("two dependent fma chains");
    time_fill<6, 5>("two dependent fma chains");
    time_fill<8, 5>("two dependent fma chains");
    time_fill<6, 6>("four dependent fma chains");
    time_fill<8, 6>("four dependent fma chains");
    time_fill<4, 3>("exp2 : fma 1 : 3");
    time_fill<8, 3>("exp2 : fma 1 : 3");
    return 0;
}
