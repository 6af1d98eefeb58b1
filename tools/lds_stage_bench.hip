// Micro-benchmark of the K6 staging pattern: 8 waves transpose their D-layout tiles (lane (m, q) holds features
// 16 t + 4 q + r of sample 16 w + m) into an LDS image.  Variants: feature-major b32 writes with row stride LD
// (T[feature][sample]) and sample-major b128 writes (T[sample][feature]).  Prints cycles per 8-tile staging.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_stage_bench tools/lds_stage_bench.hip && tools/bin/lds_stage_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LD, int MODE>
__global__ __launch_bounds__(512) void k(long long *out, float *sink, int reps)
{
    extern __shared__ __attribute__((aligned(16))) float T[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, q = lane >> 4, col = 16 * wave + m;
    f32x4 a[8];
    for (int t = 0; t < 8; ++t) a[t] = f32x4{(float)lane, (float)t, (float)wave, 1.f};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < reps; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) T[(16 * t + 4 * q + r) * LD + col] = a[t][r];
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4 *>(T + col * LD + 16 * t + 4 * q) = a[t];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[wave] = (long long)(t1 - t0);
    if (threadIdx.x == 0) sink[0] = T[5];
}

template <int LD, int MODE>
void run(const char *name)
{
    long long *d; float *s;
    hipMalloc(&d, 64); hipMalloc(&s, 64);
    const int reps = 100;
    const size_t lds = (size_t)128 * LD * 4 + 64;
    hipFuncSetAttribute((const void *)k<LD, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LD, MODE>), dim3(1), dim3(512), lds, 0, d, s, reps);
    long long h[8];
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("%-34s LD=%3d  %7.1f ticks per staging (wave 0), %7.1f (wave 7)\n", name, LD, (double)h[0] / reps, (double)h[7] / reps);
    hipFree(d); hipFree(s);
}

int main()
{
    run<129, 0>("feature-major b32");
    run<130, 0>("feature-major b32");
    run<132, 0>("feature-major b32");
    run<136, 0>("feature-major b32");
    run<144, 0>("feature-major b32");
    run<132, 1>("sample-major b128");
    run<136, 1>("sample-major b128");
    run<140, 1>("sample-major b128");
    return 0;
}
