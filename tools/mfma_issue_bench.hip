// Micro-benchmark: what one wave alone on its SIMD pays per v_mfma_f32_32x32x2_f32 (64-cycle fp32 MFMA) depending on
// what sits between consecutive MFMAs.  256 threads = 4 waves, one per SIMD, one workgroup per CU (LDS-limited), whole
// chip busy.  Prints s_memtime ticks per MFMA for each instruction pattern (hand-placed in inline asm: the compiler
// cannot reorder it).  Background for the instruction-stream rules of ppo_step_w4.hip.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_issue_bench tools/mfma_issue_bench.hip && tools/bin/mfma_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define M0 "v_mfma_f32_32x32x2_f32 %[c0], %[a], %[b], %[c0]\n\t"
#define M1 "v_mfma_f32_32x32x2_f32 %[c1], %[a], %[b], %[c1]\n\t"
#define VI "v_fma_f32 %[x], %[z], %[z], %[x]\n\t"          /* independent-ish VALU (accumulates into %4) */
#define VJ "v_fma_f32 %[y], %[z], %[z], %[y]\n\t"
#define VD "v_fma_f32 %[x], %[x], %[z], %[z]\n\t"          /* dependent chain through %4 */
#define VX "v_exp_f32 %[y], %[z]\n\t"                  /* transcendental */
#define NOP "s_nop 0\n\t"
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)

// MODE: see names[] in main
template <int MODE>
__global__ __launch_bounds__(256) void k(long long *out, float *sink, float *gbuf, int reps)
{
    extern __shared__ __attribute__((aligned(16))) float T[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc0 = {0}, acc1 = {0};
    float a = 1.0f + lane * 1e-3f, b = 0.5f, x = 0.1f, y = 0.2f, z = 1e-3f * lane;
    for (int i = threadIdx.x; i < 8192; i += 256) T[i] = (float)i;
    __syncthreads();
    const unsigned lds_off = 16u * lane;       // byte offset inside the dynamic LDS block (it starts at 0)
    f32x4 l0 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
    float wv = 3.f, wv2 = 4.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pk0 = {x, y}, pk1 = {y, x}, pk2 = {1.0f, 0.999f};
    const unsigned bc_off = 16u * (lane >> 5);
    float *g1 = gbuf + (size_t)blockIdx.x * 16384 + 1024 * wave + lane;                 // lane-contiguous dwords
    float *g4 = gbuf + (size_t)blockIdx.x * 16384 + 1024 * wave + 4 * lane;             // lane-contiguous 16-byte units
    float *gs = gbuf + (size_t)blockIdx.x * 16384 + 32 * threadIdx.x;            // one 128-byte line per lane
    unsigned qaddr = 0;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < reps; ++it) {
        if (MODE == 0)        // 16 dependent MFMAs, one chain
            asm volatile(R8(M0) R8(M0) : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b));
        else if (MODE == 1)   // two alternating chains
            asm volatile(R8(M0 M1) : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b));
        else if (MODE == 2)   // one chain, 1 s_nop between dependent MFMAs
            asm volatile(R8(M0 NOP) R8(M0 NOP) : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b));
        else if (MODE == 3)   // one chain, 4 independent VALU between dependent MFMAs
            asm volatile(R8(M0 VI VJ VI VJ) R8(M0 VI VJ VI VJ) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 4)   // two chains, 4 independent VALU per gap
            asm volatile(R8(M0 VI VJ VI VJ M1 VI VJ VI VJ) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 5)   // two chains, 8 independent VALU per gap
            asm volatile(R8(M0 R4(VI VJ) M1 R4(VI VJ)) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 6)   // two chains, 12 independent VALU per gap
            asm volatile(R8(M0 R4(VI VJ VI) M1 R4(VI VJ VI)) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 7)   // two chains, 16 independent VALU per gap
            asm volatile(R8(M0 R8(VI VJ) M1 R8(VI VJ)) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 8)   // two chains, 8 DEPENDENT VALU per gap
            asm volatile(R8(M0 R8(VD) M1 R8(VD)) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 9)   // two chains, 6 independent VALU + 2 transcendentals per gap
            asm volatile(R8(M0 VI VJ VI VX VI VJ VI VX M1 VI VJ VI VX VI VJ VI VX) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 10)  // two chains, accumulators in VGPRs instead of AGPRs
            asm volatile(R8(M0 M1) : [c0] "+v"(acc0), [c1] "+v"(acc1) : [a] "v"(a), [b] "v"(b));
        else if (MODE == 11) {  // two chains, two ds_read_b128 + a wait per 8 MFMAs (the layer loop's operand traffic)
            asm volatile("ds_read_b128 %[l0], %[p]\n\tds_read_b128 %[l1], %[p] offset:2048\n\t" R4(M0 M1) "s_waitcnt lgkmcnt(0)\n\t"
                         "ds_read_b128 %[l0], %[p] offset:4096\n\tds_read_b128 %[l1], %[p] offset:6144\n\t" R4(M0 M1) "s_waitcnt lgkmcnt(0)\n\t"
                         : [c0] "+a"(acc0), [c1] "+a"(acc1), [l0] "=&v"(l0), [l1] "=&v"(l1) : [a] "v"(a), [b] "v"(b), [p] "v"(lds_off));
        } else if (MODE == 12)  // two chains, 8 ds_read_b32 per 8 MFMAs
            asm volatile(R8(M0 "ds_read_b32 %[x], %[p]\n\t" M1 "s_waitcnt lgkmcnt(4)\n\t") : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "=&v"(x) : [a] "v"(a), [b] "v"(b), [p] "v"(lds_off));
        else if (MODE == 13)  // two chains, 1 s_nop per gap
            asm volatile(R8(M0 NOP M1 NOP) : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b));
        else if (MODE == 14)  // two chains, v_accvgpr_read + write (of another AGPR) per gap
            asm volatile(R8(M0 "v_accvgpr_read_b32 %[x], %[w]\n\tv_accvgpr_write_b32 %[w], %[x]\n\t" M1 "v_accvgpr_read_b32 %[y], %[w]\n\tv_accvgpr_write_b32 %[w], %[y]\n\t")
                         : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y), [w] "+a"(wv) : [a] "v"(a), [b] "v"(b));
        else if (MODE == 20)  // VALU only: 16 v_fma_f32 on two independent registers
            asm volatile(R8(VI VJ) : [x] "+v"(x), [y] "+v"(y) : [z] "v"(z));
        else if (MODE == 21)  // 16 dependent v_fma_f32
            asm volatile(R8(VD VD) : [x] "+v"(x) : [z] "v"(z));
        else if (MODE == 22)  // 16 v_pk_fma_f32 on two independent register pairs
            asm volatile(R8("v_pk_fma_f32 %[p], %[r], %[r], %[p]\n\tv_pk_fma_f32 %[q], %[r], %[r], %[q]\n\t") : [p] "+v"(pk0), [q] "+v"(pk1) : [r] "v"(pk2));
        else if (MODE == 23)  // 16 dependent v_pk_fma_f32 with the s_nop hipcc puts between them
            asm volatile(R8("v_pk_fma_f32 %[p], %[p], %[r], %[r]\n\ts_nop 0\n\tv_pk_fma_f32 %[p], %[p], %[r], %[r]\n\ts_nop 0\n\t") : [p] "+v"(pk0) : [r] "v"(pk2));
        else if (MODE == 24)  // 16 dependent v_pk_fma_f32, no s_nop (timing only)
            asm volatile(R8("v_pk_fma_f32 %[p], %[p], %[r], %[r]\n\tv_pk_fma_f32 %[p], %[p], %[r], %[r]\n\t") : [p] "+v"(pk0) : [r] "v"(pk2));
        else if (MODE == 25)  // 16 v_exp_f32 (independent sources)
            asm volatile(R8("v_exp_f32 %[x], %[z]\n\tv_exp_f32 %[y], %[z]\n\t") : [x] "+v"(x), [y] "+v"(y) : [z] "v"(z));
        else if (MODE == 26)  // 16 v_rcp_f32
            asm volatile(R8("v_rcp_f32 %[x], %[z]\n\tv_rcp_f32 %[y], %[z]\n\t") : [x] "+v"(x), [y] "+v"(y) : [z] "v"(z));
        else if (MODE == 27)  // 16 v_accvgpr_write_b32 (two AGPRs)
            asm volatile(R8("v_accvgpr_write_b32 %[w], %[z]\n\tv_accvgpr_write_b32 %[w2], %[z]\n\t") : [w] "+a"(wv), [w2] "+a"(wv2) : [z] "v"(z));
        else if (MODE == 28)  // 16 v_accvgpr_read_b32
            asm volatile(R8("v_accvgpr_read_b32 %[x], %[w]\n\tv_accvgpr_read_b32 %[y], %[w2]\n\t") : [x] "+v"(x), [y] "+v"(y), [w] "+a"(wv), [w2] "+a"(wv2));
        else if (MODE == 29)  // 16 v_bfi_b32
            asm volatile(R8("v_bfi_b32 %[x], %[z], %[x], %[y]\n\tv_bfi_b32 %[y], %[z], %[y], %[x]\n\t") : [x] "+v"(x), [y] "+v"(y) : [z] "v"(z));
        else if (MODE == 30)  // 16 independent v_fma with 16 s_nop 0 interleaved
            asm volatile(R8(VI NOP VJ NOP) : [x] "+v"(x), [y] "+v"(y) : [z] "v"(z));
        else if (MODE == 31)  // 16 v_mov_b32
            asm volatile(R8("v_mov_b32 %[x], %[z]\n\tv_mov_b32 %[y], %[z]\n\t") : [x] "+v"(x), [y] "+v"(y) : [z] "v"(z));
        else if (MODE == 32)  // 8 x (ds_read_b128 broadcast-per-half + wait)
            asm volatile(R8("ds_read_b128 %[l0], %[p]\n\ts_waitcnt lgkmcnt(0)\n\t") R8("ds_read_b128 %[l0], %[p]\n\ts_waitcnt lgkmcnt(0)\n\t") : [l0] "=&v"(l0) : [p] "v"(bc_off));
        else if (MODE == 33)  // 16 x (ds_read_b128 distinct addresses + wait)
            asm volatile(R8("ds_read_b128 %[l0], %[p]\n\ts_waitcnt lgkmcnt(0)\n\t") R8("ds_read_b128 %[l0], %[p]\n\ts_waitcnt lgkmcnt(0)\n\t") : [l0] "=&v"(l0) : [p] "v"(lds_off));
        else if (MODE == 49 || MODE == 50) {  // the dW loop as hipcc builds it: the 4 MFMAs of a group take their operands from two
                                // 16-byte LDS reads issued one group earlier (50: operands are loop constants, reads still issued)
            const float *pa = T + 4 * lane + 64 * (it & 1);
            float4 av[2], bv[2];
            av[0] = *reinterpret_cast<const float4 *>(pa);
            bv[0] = *reinterpret_cast<const float4 *>(pa + 512);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                av[(j + 1) & 1] = *reinterpret_cast<const float4 *>(pa + 1024 * ((j + 1) & 3));
                bv[(j + 1) & 1] = *reinterpret_cast<const float4 *>(pa + 1024 * ((j + 1) & 3) + 512);
                const float4 u = av[j & 1], v = bv[j & 1];
                if (MODE == 49) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u.x, v.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u.y, v.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u.z, v.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u.w, v.w, acc0, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    x += u.x + v.y;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        else if (MODE == 51)  // two chains, 1 VALU per gap
            asm volatile(R8(M0 VI M1 VJ) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 52)  // two chains, 1 VALU per two MFMAs
            asm volatile(R8(M0 VI M1) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 53)  // two chains, 2 VALU per gap
            asm volatile(R8(M0 VI VJ M1 VI VJ) : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        else if (MODE == 54)  // one chain, 1 VALU + 1 ds_read2_b32 per gap (the transposed-operand loop of the backward layer)
            asm volatile(R8(M0 "v_add_u32 %[q], 0x400, %[p]\n\tds_read2_b32 %[l2], %[q] offset0:8 offset1:140\n\t" M0 "s_waitcnt lgkmcnt(0)\n\t")
                         : [c0] "+a"(acc0), [c1] "+a"(acc1), [q] "=&v"(qaddr), [l2] "=&v"(pk0) : [a] "v"(a), [b] "v"(b), [p] "v"(lds_off));
        else if (MODE == 40)  // two chains + one coalesced global_store_dword per gap (data in a VGPR)
            asm volatile(M0 "global_store_dword %[g1], %[x], off offset:0\n\t" M1 "global_store_dword %[g1], %[y], off offset:256\n\t" M0 "global_store_dword %[g1], %[x], off offset:512\n\t" M1 "global_store_dword %[g1], %[y], off offset:768\n\t" M0 "global_store_dword %[g1], %[x], off offset:1024\n\t" M1 "global_store_dword %[g1], %[y], off offset:1280\n\t" M0 "global_store_dword %[g1], %[x], off offset:1536\n\t" M1 "global_store_dword %[g1], %[y], off offset:1792\n\t" M0 "global_store_dword %[g1], %[x], off offset:2048\n\t" M1 "global_store_dword %[g1], %[y], off offset:2304\n\t" M0 "global_store_dword %[g1], %[x], off offset:2560\n\t" M1 "global_store_dword %[g1], %[y], off offset:2816\n\t" M0 "global_store_dword %[g1], %[x], off offset:3072\n\t" M1 "global_store_dword %[g1], %[y], off offset:3328\n\t" M0 "global_store_dword %[g1], %[x], off offset:3584\n\t" M1 "global_store_dword %[g1], %[y], off offset:3840\n\t"
                         : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y), [g1] "v"(g1) : "memory");
        else if (MODE == 41)  // two chains + one coalesced global_store_dwordx4 per 4 MFMAs (same bytes as 40)
            asm volatile(M0 M1 M0 "global_store_dwordx4 %[g4], %[l0], off offset:0\n\t" M1 M0 M1 M0 "global_store_dwordx4 %[g4], %[l0], off offset:1024\n\t" M1 M0 M1 M0 "global_store_dwordx4 %[g4], %[l0], off offset:2048\n\t" M1 M0 M1 M0 "global_store_dwordx4 %[g4], %[l0], off offset:3072\n\t" M1
                         : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b), [l0] "v"(l0), [g4] "v"(g4) : "memory");
        else if (MODE == 42)  // as 40 with the store data in an AGPR
            asm volatile(M0 "global_store_dword %[g1], %[w], off offset:0\n\t" M1 "global_store_dword %[g1], %[w2], off offset:256\n\t" M0 "global_store_dword %[g1], %[w], off offset:512\n\t" M1 "global_store_dword %[g1], %[w2], off offset:768\n\t" M0 "global_store_dword %[g1], %[w], off offset:1024\n\t" M1 "global_store_dword %[g1], %[w2], off offset:1280\n\t" M0 "global_store_dword %[g1], %[w], off offset:1536\n\t" M1 "global_store_dword %[g1], %[w2], off offset:1792\n\t" M0 "global_store_dword %[g1], %[w], off offset:2048\n\t" M1 "global_store_dword %[g1], %[w2], off offset:2304\n\t" M0 "global_store_dword %[g1], %[w], off offset:2560\n\t" M1 "global_store_dword %[g1], %[w2], off offset:2816\n\t" M0 "global_store_dword %[g1], %[w], off offset:3072\n\t" M1 "global_store_dword %[g1], %[w2], off offset:3328\n\t" M0 "global_store_dword %[g1], %[w], off offset:3584\n\t" M1 "global_store_dword %[g1], %[w2], off offset:3840\n\t"
                         : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b), [w] "a"(wv), [w2] "a"(wv2), [g1] "v"(g1) : "memory");
        else if (MODE == 43)  // stores only: 16 coalesced global_store_dword
            asm volatile("global_store_dword %[g1], %[x], off offset:0\n\t" "global_store_dword %[g1], %[y], off offset:256\n\t" "global_store_dword %[g1], %[x], off offset:512\n\t" "global_store_dword %[g1], %[y], off offset:768\n\t" "global_store_dword %[g1], %[x], off offset:1024\n\t" "global_store_dword %[g1], %[y], off offset:1280\n\t" "global_store_dword %[g1], %[x], off offset:1536\n\t" "global_store_dword %[g1], %[y], off offset:1792\n\t" "global_store_dword %[g1], %[x], off offset:2048\n\t" "global_store_dword %[g1], %[y], off offset:2304\n\t" "global_store_dword %[g1], %[x], off offset:2560\n\t" "global_store_dword %[g1], %[y], off offset:2816\n\t" "global_store_dword %[g1], %[x], off offset:3072\n\t" "global_store_dword %[g1], %[y], off offset:3328\n\t" "global_store_dword %[g1], %[x], off offset:3584\n\t" "global_store_dword %[g1], %[y], off offset:3840\n\t"
                         : : [x] "v"(x), [y] "v"(y), [g1] "v"(g1) : "memory");
        else if (MODE == 44)  // 16 MFMAs, then the 16 stores in one burst
            asm volatile(R8(M0 M1) "global_store_dword %[g1], %[x], off offset:0\n\t" "global_store_dword %[g1], %[y], off offset:256\n\t" "global_store_dword %[g1], %[x], off offset:512\n\t" "global_store_dword %[g1], %[y], off offset:768\n\t" "global_store_dword %[g1], %[x], off offset:1024\n\t" "global_store_dword %[g1], %[y], off offset:1280\n\t" "global_store_dword %[g1], %[x], off offset:1536\n\t" "global_store_dword %[g1], %[y], off offset:1792\n\t" "global_store_dword %[g1], %[x], off offset:2048\n\t" "global_store_dword %[g1], %[y], off offset:2304\n\t" "global_store_dword %[g1], %[x], off offset:2560\n\t" "global_store_dword %[g1], %[y], off offset:2816\n\t" "global_store_dword %[g1], %[x], off offset:3072\n\t" "global_store_dword %[g1], %[y], off offset:3328\n\t" "global_store_dword %[g1], %[x], off offset:3584\n\t" "global_store_dword %[g1], %[y], off offset:3840\n\t"
                         : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y), [g1] "v"(g1) : "memory");
        else if (MODE == 45)  // two chains + one ds_write_b32 per gap
            asm volatile(R8(M0 "ds_write_b32 %[p], %[x]\n\t" M1 "ds_write_b32 %[p], %[y] offset:1024\n\t")
                         : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y), [p] "v"(lds_off) : "memory");
        else if (MODE == 46)  // two chains + one ds_write_b128 per 4 MFMAs
            asm volatile(R4(M0 M1 M0 "ds_write_b128 %[p], %[l0]\n\t" M1)
                         : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b), [l0] "v"(l0), [p] "v"(lds_off) : "memory");
        else if (MODE == 47)  // two chains + one global_store_dword per gap, 64 lanes on 64 different 128-byte lines
            asm volatile(R8(M0 "global_store_dword %[gs], %[x], off\n\t" M1 "global_store_dword %[gs], %[y], off offset:4\n\t")
                         : [c0] "+a"(acc0), [c1] "+a"(acc1) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y), [gs] "v"(gs) : "memory");
        else if (MODE == 48)  // two chains + one global_load_dword per gap (coalesced), waited at the end of the 16
            asm volatile(R8(M0 "global_load_dword %[x], %[g1], off\n\t" M1 "global_load_dword %[y], %[g1], off offset:256\n\t") "s_waitcnt vmcnt(0)\n\t"
                         : [c0] "+a"(acc0), [c1] "+a"(acc1), [x] "=&v"(x), [y] "=&v"(y) : [a] "v"(a), [b] "v"(b), [g1] "v"(g1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0 && blockIdx.x == 0) out[wave] = (long long)(t1 - t0);
    float s = x + y + (float)qaddr + l0[0] + l1[1] + wv + wv2 + pk0.x + pk1.y;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int MODE>
void run(const char *name)
{
    long long *d; float *s;
    static float *gbuf = nullptr;
    if (!gbuf) hipMalloc(&gbuf, (size_t)256 * 16384 * 4);
    hipMalloc(&d, 64); hipMalloc(&s, 4096);
    const int reps = 200;
    const size_t lds = 100 * 1024;   // one workgroup per CU
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int i = 0; i < 3; ++i) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), lds, 0, d, s, gbuf, reps);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long h[4];
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (reps * 16.0);
    printf("%-62s %7.1f ticks / MFMA   (kernel %.1f us => %.2f GHz if ticks are shader cycles)\n", name, per, ms * 1e3,
           (double)h[0] / (ms * 1e-3) * 1e-9);
    hipFree(d); hipFree(s);
}

int main()
{
    run<0>("0  one chain (dependent, back-to-back)");
    run<1>("1  two alternating chains");
    run<2>("2  one chain + 1 s_nop between dependent MFMAs");
    run<3>("3  one chain + 4 indep VALU between dependent MFMAs");
    run<4>("4  two chains + 4 indep VALU per gap");
    run<5>("5  two chains + 8 indep VALU per gap");
    run<6>("6  two chains + 12 indep VALU per gap");
    run<7>("7  two chains + 16 indep VALU per gap");
    run<8>("8  two chains + 8 DEPENDENT VALU per gap");
    run<9>("9  two chains + 6 VALU + 2 v_exp per gap");
    run<10>("10 two chains, accumulators in arch VGPRs");
    run<11>("11 two chains + 2 ds_read_b128 + wait per 8 MFMAs");
    run<12>("12 two chains + ds_read_b32 per 2 MFMAs");
    run<13>("13 two chains + 1 s_nop per gap");
    run<14>("14 two chains + accvgpr read+write per gap");
    run<49>("49 dW-loop shape: operands from ds_read_b128 a group ahead (hipcc)");
    run<50>("50 as 49 with constant MFMA operands (+2 VALU per group)");
    run<51>("51 two chains + 1 VALU per gap");
    run<52>("52 two chains + 1 VALU per two MFMAs");
    run<53>("53 two chains + 2 VALU per gap");
    run<54>("54 one chain + (v_add + ds_read2_b32) per two MFMAs");
    run<40>("40 two chains + coalesced global_store_dword per gap");
    run<41>("41 two chains + coalesced global_store_dwordx4 per 4 MFMAs");
    run<42>("42 as 40, store data in AGPRs");
    run<44>("44 16 MFMAs then 16 global_store_dword in a burst");
    run<45>("45 two chains + ds_write_b32 per gap");
    run<46>("46 two chains + ds_write_b128 per 4 MFMAs");
    run<47>("47 two chains + global_store_dword per gap, one line per lane");
    run<48>("48 two chains + coalesced global_load_dword per gap");
    printf("--- VALU only: ticks per 16 instructions are (ticks / 'MFMA') x 1, i.e. per instruction = value / 16 ...\n");
    run<20>("20 16 v_fma_f32, two independent accumulators      [per 16]");
    run<21>("21 16 v_fma_f32, one dependent chain               [per 16]");
    run<22>("22 16 v_pk_fma_f32, two independent                [per 16]");
    run<23>("23 16 v_pk_fma_f32 dependent + s_nop 0 each        [per 16]");
    run<24>("24 16 v_pk_fma_f32 dependent, no s_nop             [per 16]");
    run<25>("25 16 v_exp_f32                                    [per 16]");
    run<26>("26 16 v_rcp_f32                                    [per 16]");
    run<27>("27 16 v_accvgpr_write_b32                          [per 16]");
    run<28>("28 16 v_accvgpr_read_b32                           [per 16]");
    run<29>("29 16 v_bfi_b32 (dependent pair)                   [per 16]");
    run<30>("30 16 v_fma_f32 + 16 s_nop 0                       [per 16]");
    run<31>("31 16 v_mov_b32                                    [per 16]");
    run<43>("43 16 global_store_dword, coalesced                [per 16]");
    run<32>("32 16 x (ds_read_b128 half-broadcast + wait)       [per 16]");
    run<33>("33 16 x (ds_read_b128 distinct + wait)             [per 16]");
    return 0;
}
