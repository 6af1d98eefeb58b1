// fp32 MFMA GEMMs of the layered path (any build_mlp shape), with the epilogues that used to be separate launches.
// gfx950, v_mfma_f32_32x32x2_f32 (exact fp32).  One kernel template covers the three contractions of a dense layer:
//
//   forward   Y[m][n]  = act( sum_k X[m][k]  W[n][k] + b[n] )           A = X  (RC)   B = W  (RC)   epilogue BIAS / BIAS_GELU
//   backward  dH[m][c] = ( sum_n dZ[m][n] W[n][c] ) * GELU'[m][c]        A = dZ (RC)   B = W  (OC)   epilogue MUL / STORE / ACC
//   weights   dW[n][c] = sum_m dZ[m][n] X[m][c],  db[n] = sum_m dZ[m][n] A = dZ (OC)   B = X  (OC)   epilogue PARTIAL
//
// C[i][j] = sum_k A(i, k) B(j, k).  An operand is "RC" when its reduction index is the contiguous one in memory
// (elem(i, k) = P[i ld + k]) and "OC" when its output index is (elem(i, k) = P[k ld + i]).  A workgroup of 4 waves
// owns a 64 x 64 tile of C (each wave one 32 x 32 accumulator); the reduction advances in chunks of 32 through double-buffered
// LDS tiles stored reduction-major (T[k][i], row stride 68), so an MFMA operand read is 32 consecutive floats per lane
// half; the next chunks' global loads are in flight under the current chunk's MFMAs.  Shapes are arbitrary: loads are
// clamped and masked, stores bounds-checked.  The weight-gradient contraction runs over the batch (tens of thousands
// of rows) into a small output: it is split over blockIdx.z in chunks of rows, each split writes its own partial
// (fixed-order sum afterwards: deterministic), and the workgroups of the first column tile also emit the row sums of
// A -- the bias gradient -- from the LDS tiles they already hold.
#pragma once
#include <stdlib.h>

#include "mlp_chain.h"

namespace {

constexpr int GT = 64;    // C tile edge
constexpr int GK = 32;    // reduction chunk
constexpr int GLD = 68;   // LDS row stride (floats)

enum { OP_RC = 0, OP_OC = 1 };
enum { EPI_STORE = 0, EPI_ACC, EPI_BIAS, EPI_BIAS_GELU, EPI_MUL, EPI_PARTIAL };

struct GemmArgs {
    const float *A, *B;
    int lda, ldb;
    int M, N, K;              // C is M x N; K = reduction length
    float *C;
    int ldc;
    const float *bias;        // BIAS / BIAS_GELU: [N]
    float *G;                 // BIAS_GELU: GELU'(z) out, same layout as C (may be NULL); MUL: gate in
    int kchunk;               // PARTIAL: reduction rows per blockIdx.z
    int64_t c_split;          // PARTIAL: floats between consecutive partial C's
    float *rowsum;            // PARTIAL: sum_k A(i, k) of the split (may be NULL)
    int64_t rs_split;         // PARTIAL: floats between consecutive partial row-sum vectors
    // batched launch (gemm_small_kernel only: the SAC critic ensemble's E same-shaped layers in one grid): problem b uses
    // A + b sA, B + b sB, C + b sC, bias + b sBias, G + b sG, rowsum + b sRS; blockIdx.z = b * nsplit + split
    int nbatch, nsplit;
    int64_t sA, sB, sC, sBias, sG, sRS;
};

// A thread's eight elements (two groups of four) of a 64 x 32 operand tile.  VEC (16-byte aligned base, ld % 4 == 0 and the
// contiguous extent a multiple of 4): two 16-byte loads along the contiguous index; otherwise eight clamped scalar loads.
template <int OP, bool VEC>
__device__ __forceinline__ void tile_load(float (&r)[8], const float *__restrict__ P, int ld, int i0, int imax, int k0, int kmax, int tid)
{
    if (VEC) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = OP == OP_RC ? tid >> 2 : (tid & 15) * 4, k = OP == OP_RC ? ((tid & 3) + 4 * h) * 4 : (tid >> 4) + 16 * h;
            const int gi = i0 + i, gk = k0 + k;
            const bool ok = gi < imax && gk < kmax;  // the extent along the vector is a multiple of 4: all four or none
            const size_t off = !ok ? 0 : (OP == OP_RC ? (size_t)gi * ld + gk : (size_t)gk * ld + gi);
            const float4 v = *reinterpret_cast<const float4 *>(P + off);
            r[4 * h + 0] = ok ? v.x : 0.f; r[4 * h + 1] = ok ? v.y : 0.f; r[4 * h + 2] = ok ? v.z : 0.f; r[4 * h + 3] = ok ? v.w : 0.f;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            const int i = OP == OP_RC ? e >> 5 : e & 63, k = OP == OP_RC ? e & 31 : e >> 6;
            const int gi = i0 + i, gk = k0 + k;
            const bool ok = gi < imax && gk < kmax;
            const size_t off = !ok ? 0 : (OP == OP_RC ? (size_t)gi * ld + gk : (size_t)gk * ld + gi);
            const float v = P[off];
            r[u] = ok ? v : 0.f;
        }
    }
}

template <int OP, bool VEC, int LD = GLD>      // LD: row stride of the reduction-major LDS tile T[k][i]
__device__ __forceinline__ void tile_store(const float (&r)[8], float *T, int tid)
{
    if (VEC && OP == OP_OC) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            *reinterpret_cast<float4 *>(T + ((tid >> 4) + 16 * h) * LD + (tid & 15) * 4) = make_float4(r[4 * h], r[4 * h + 1], r[4 * h + 2], r[4 * h + 3]);
    } else if (VEC) {
        const int i = tid >> 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = ((tid & 3) + 4 * h) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) T[(k + c) * LD + i] = r[4 * h + c];
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            const int i = OP == OP_RC ? e >> 5 : e & 63, k = OP == OP_RC ? e & 31 : e >> 6;
            T[k * LD + i] = r[u];
        }
    }
}

// Main loop: the reduction advances in chunks of GK = 32 through DOUBLE-BUFFERED LDS tiles -- one workgroup barrier per chunk
// (16 MFMAs per wave and accumulator between barriers; the first version had two barriers per 8 MFMAs and ran the forward layers at
// 20-40 TFLOP/s) -- with the global loads of the two following chunks in flight in registers:
//   iteration c:  regs(c + 1) -> LDS[(c + 1) & 1];  issue loads of chunk c + 3 into that register slot;  MFMAs on LDS[c & 1];  barrier.
// Tile shape (round 3): (64 RM) x (64 RN), RM, RN in {1, 2}; a wave owns RM x RN accumulators of 32 x 32.  With 64 x 64 tiles a
// workgroup needs 16 KB of operands per 1024 MFMA cycles = 16 B/clk, and a CU streams ~12 B/clk from beyond its (cold) L2 (measured
// in the fused SAC kernels): the GEMMs ran at 20-33 % of the fp32 MFMA peak whatever their FLOP count.
// 128 x 128 tiles need 8 B/clk (and read the LDS operands once per two MFMAs instead of once per MFMA) -- built to test that
// explanation, and it is wrong for these GEMMs: every larger shape measured slower (see gemm_launch); 64 x 64 stays the default.
template <int AOP, int BOP, int EPI, bool VA, bool VB, int RM, int RN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g)
{
    constexpr int TM = GT * RM, TN = GT * RN, LDA = TM + 4, LDB = TN + 4;
    extern __shared__ __attribute__((aligned(16))) float gemm_lds[];
    float *As0 = gemm_lds, *As1 = As0 + GK * LDA, *Bs0 = As1 + GK * LDA, *Bs1 = Bs0 + GK * LDB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (linear id mod 8), each with its own L2: the nx column
    // tiles of one row tile all read the same rows of A (the activations: the big operand), so they should share an L2 and run
    // at the same time.  Linear id L -> xcd = L % 8, slot = L / 8; row tile = (slot / nx) * 8 + xcd, column tile = slot % nx: the nx
    // tiles of a row tile are nx consecutive slots of ONE XCD.  (Rows beyond the last full group of 8 keep the plain order.)
    int ty = blockIdx.y, tx = blockIdx.x;
    {
        const int nx = gridDim.x, ny = gridDim.y, lin = blockIdx.y * nx + blockIdx.x, full = (ny / 8) * 8 * nx;
        if (lin < full) {
            const int xcd = lin & 7, slot = lin >> 3;
            ty = (slot / nx) * 8 + xcd;
            tx = slot - (slot / nx) * nx;
        }
    }
    const int i0 = ty * TM, j0 = tx * TN;
    int kb = 0, ke = g.K;
    if (EPI == EPI_PARTIAL) {
        kb = blockIdx.z * g.kchunk;
        ke = min(g.K, kb + g.kchunk);
    }
    const bool want_rs = EPI == EPI_PARTIAL && g.rowsum != nullptr && tx == 0;
    const int nch = (ke - kb + GK - 1) / GK;
    float ra[2][RM][8], rb[2][RN][8], rsum = 0.f;
    f32x16 acc[RM][RN];
#pragma unroll
    for (int a = 0; a < RM; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b) acc[a][b] = f32x16{0};
    auto load = [&](int slot, int k0) {
#pragma unroll
        for (int a = 0; a < RM; ++a) tile_load<AOP, VA>(ra[slot][a], g.A, g.lda, i0 + GT * a, g.M, k0, ke, tid);
#pragma unroll
        for (int b = 0; b < RN; ++b) tile_load<BOP, VB>(rb[slot][b], g.B, g.ldb, j0 + GT * b, g.N, k0, ke, tid);
    };
    auto store = [&](int slot, float *As, float *Bs) {
#pragma unroll
        for (int a = 0; a < RM; ++a) tile_store<AOP, VA, LDA>(ra[slot][a], As + GT * a, tid);
#pragma unroll
        for (int b = 0; b < RN; ++b) tile_store<BOP, VB, LDB>(rb[slot][b], Bs + GT * b, tid);
    };
    // chunk 0 straight to LDS[0]; chunks 1 and 2 into the register slots
    load(0, kb);
    load(1, kb + GK);
    store(0, As0, Bs0);
    load(0, kb + 2 * GK);
    __syncthreads();
    for (int c = 0; c < nch; c += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {                          // chunk c + d lives in LDS[d]; register slot (d + 1) & 1 holds chunk c + d + 1
            const int cc = c + d;
            if (cc < nch) {                                    // uniform
                const int slot = (d + 1) & 1;
                float *Asd = d ? As1 : As0, *Bsd = d ? Bs1 : Bs0;
                if (cc + 1 < nch) {
                    store(slot, slot ? As1 : As0, slot ? Bs1 : Bs0);
                    if (cc + 3 < nch) load(slot, kb + (cc + 3) * GK);
                }
                const float *a = Asd + hi * LDA + 32 * RM * wm + l31, *b = Bsd + hi * LDB + 32 * RN * wn + l31;
#pragma unroll
                for (int s2 = 0; s2 < GK / 2; ++s2) {
                    float av[RM], bv[RN];
#pragma unroll
                    for (int x = 0; x < RM; ++x) av[x] = a[2 * s2 * LDA + 32 * x];
#pragma unroll
                    for (int y = 0; y < RN; ++y) bv[y] = b[2 * s2 * LDB + 32 * y];
#pragma unroll
                    for (int x = 0; x < RM; ++x)
#pragma unroll
                        for (int y = 0; y < RN; ++y) acc[x][y] = mfma32(av[x], bv[y], acc[x][y]);
                }
                if (want_rs && tid < TM) {
#pragma unroll
                    for (int k = 0; k < GK; ++k) rsum += Asd[k * LDA + tid];
                }
                __syncthreads();                               // LDS[slot] is complete, LDS[d] is free for chunk cc + 2
            }
        }
    }

    float *C = g.C + (EPI == EPI_PARTIAL ? (size_t)blockIdx.z * g.c_split : 0);
#pragma unroll
    for (int y = 0; y < RN; ++y) {
        const int j = j0 + 32 * (RN * wn + y) + l31;
        if (j >= g.N) continue;
        const float bj = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? g.bias[j] : 0.f;
#pragma unroll
        for (int x = 0; x < RM; ++x) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + 32 * (RM * wm + x) + crow(r, hi);
                if (i < g.M) {
                    const size_t o = (size_t)i * g.ldc + j;
                    const float v = acc[x][y][r];
                    if (EPI == EPI_STORE || EPI == EPI_PARTIAL) C[o] = v;
                    else if (EPI == EPI_ACC) C[o] += v;
                    else if (EPI == EPI_BIAS) C[o] = v + bj;
                    else if (EPI == EPI_MUL) C[o] = v * g.G[o];
                    else {
                        float yv, gd;
                        gelu_and_grad_fast(v + bj, yv, gd);
                        C[o] = yv;
                        if (g.G) g.G[o] = gd;
                    }
                }
            }
        }
    }
    if (want_rs && tid < TM && i0 + tid < g.M) g.rowsum[(size_t)blockIdx.z * g.rs_split + i0 + tid] = rsum;
}

// Small outputs (the off-policy agents' batches of a few hundred rows: 16 tiles of 64 x 64 would leave 240 CUs idle and
// each of the 16 walk the whole reduction): a workgroup owns a 32 x 32 tile and its 4 waves split the REDUCTION -- every
// step stages 64 reduction indices, wave w takes rows 16 w .. 16 w + 15 of the staged tiles -- and the four partial
// accumulators meet in LDS in a fixed order (deterministic), wave w finishing accumulator rows 4 w .. 4 w + 3.
constexpr int ST = 32, SK = 64, SLD = 33;   // odd stride: staging stores and operand reads both land 2 lanes per bank

template <int OP>
__device__ __forceinline__ void small_load(float (&r)[8], const float *__restrict__ P, int ld, int i0, int imax, int k0, int kmax, int tid)
{
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = tid + 256 * u;
        const int i = OP == OP_RC ? e >> 6 : e & 31, k = OP == OP_RC ? e & 63 : e >> 5;
        const int gi = i0 + i, gk = k0 + k;
        const bool ok = gi < imax && gk < kmax;
        const size_t off = !ok ? 0 : (OP == OP_RC ? (size_t)gi * ld + gk : (size_t)gk * ld + gi);
        const float v = P[off];
        r[u] = ok ? v : 0.f;
    }
}

template <int OP>
__device__ __forceinline__ void small_store(const float (&r)[8], float *T, int tid)
{
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = tid + 256 * u;
        const int i = OP == OP_RC ? e >> 6 : e & 31, k = OP == OP_RC ? e & 63 : e >> 5;
        T[k * SLD + i] = r[u];
    }
}

template <int AOP, int BOP, int EPI>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs g)
{
    __shared__ float As[SK * SLD], Bs[SK * SLD];
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int i0 = blockIdx.y * ST, j0 = blockIdx.x * ST;
    const int batch = (int)blockIdx.z / g.nsplit, split = (int)blockIdx.z - batch * g.nsplit;
    const float *gA = g.A + (size_t)batch * g.sA, *gB = g.B + (size_t)batch * g.sB;
    int kb = 0, ke = g.K;
    if (EPI == EPI_PARTIAL) {
        kb = split * g.kchunk;
        ke = min(g.K, kb + g.kchunk);
    }
    const bool want_rs = EPI == EPI_PARTIAL && g.rowsum != nullptr && blockIdx.x == 0;
    float ra[8], rb[8], rsum = 0.f;
    f32x16 acc = {0};
    small_load<AOP>(ra, gA, g.lda, i0, g.M, kb, ke, tid);
    small_load<BOP>(rb, gB, g.ldb, j0, g.N, kb, ke, tid);
    for (int k0 = kb; k0 < ke; k0 += SK) {
        __syncthreads();
        small_store<AOP>(ra, As, tid);
        small_store<BOP>(rb, Bs, tid);
        __syncthreads();
        if (k0 + SK < ke) {
            small_load<AOP>(ra, gA, g.lda, i0, g.M, k0 + SK, ke, tid);
            small_load<BOP>(rb, gB, g.ldb, j0, g.N, k0 + SK, ke, tid);
        }
        const float *a = As + (16 * wave + hi) * SLD + l31, *b = Bs + (16 * wave + hi) * SLD + l31;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = mfma32(a[2 * s * SLD], b[2 * s * SLD], acc);
        if (want_rs && tid < ST) {
#pragma unroll 16
            for (int k = 0; k < SK; ++k) rsum += As[k * SLD + tid];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();

    const int j = j0 + l31;
    float *C = g.C + (size_t)batch * g.sC + (EPI == EPI_PARTIAL ? (size_t)split * g.c_split : 0);
    float *Gm = g.G ? g.G + (size_t)batch * g.sG : nullptr;
    if (j < g.N) {
        const float bj = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? g.bias[(size_t)batch * g.sBias + j] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * wave + q;
            const int i = i0 + crow(r, hi);
            if (i < g.M) {
                const float v = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
                const size_t o = (size_t)i * g.ldc + j;
                if (EPI == EPI_STORE || EPI == EPI_PARTIAL) C[o] = v;
                else if (EPI == EPI_ACC) C[o] += v;
                else if (EPI == EPI_BIAS) C[o] = v + bj;
                else if (EPI == EPI_MUL) C[o] = v * Gm[o];
                else {
                    float y, gd;
                    gelu_and_grad_fast(v + bj, y, gd);
                    C[o] = y;
                    if (Gm) Gm[o] = gd;
                }
            }
        }
    }
    if (want_rs && tid < ST && i0 + tid < g.M) g.rowsum[(size_t)batch * g.sRS + (size_t)split * g.rs_split + i0 + tid] = rsum;
}

template <int AOP, int BOP, int EPI, bool VA, bool VB, int RM, int RN>
void gemm_go(dim3 grid, size_t lds, hipStream_t s, const GemmArgs &g)
{
    static bool attr_set = false;                 // the 128 x 128 shape stages 67.6 KB: beyond the default dynamic-LDS limit
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)gemm_kernel<AOP, BOP, EPI, VA, VB, RM, RN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<AOP, BOP, EPI, VA, VB, RM, RN>), grid, dim3(256), lds, s, g);
}

template <int AOP, int BOP, int EPI>
int gemm_launch(hipStream_t s, const GemmArgs &g_in, int splits, const char *what)
{
    GemmArgs g = g_in;
    if (g.nbatch < 1) g.nbatch = 1;
    g.nsplit = splits;
    const int64_t tiles64 = erl_cdiv(g.N, GT) * erl_cdiv(g.M, GT) * splits;
    if (tiles64 < 128 || g.nbatch > 1) {   // fewer 64 x 64 tiles than half the CUs (or a batch of small problems): 32 x 32 tiles with the
                                           // reduction split over the waves
        const dim3 grid((unsigned)erl_cdiv(g.N, ST), (unsigned)erl_cdiv(g.M, ST), (unsigned)(splits * g.nbatch));
        hipLaunchKernelGGL((gemm_small_kernel<AOP, BOP, EPI>), grid, dim3(256), 0, s, g);
    } else {
        auto vec_ok = [](int op, const float *P, int ld, int outs, int red) {
            return (reinterpret_cast<uintptr_t>(P) & 15) == 0 && ld % 4 == 0 && (op == OP_RC ? red : outs) % 4 == 0;
        };
        const bool va = vec_ok(AOP, g.A, g.lda, g.M, g.K), vb = vec_ok(BOP, g.B, g.ldb, g.N, g.K);
        // tile shape: 64 x 64.  The 128-wide shapes exist (ERL_GEMM_TILE=12|21|22 forces one) and were MEASURED SLOWER at the PPO
        // minibatch (B = 16384; profiles/r03_gemm_tile_ab.txt: whole layered step 180 / 227 / 266 us with 64 x 64 tiles against
        // 222 / 266 / 307 with 64 x 128, 227 / 264 / 308 with 128 x 64 and 329 / 361 / 418 with 128 x 128 at [128,128] / (256,128) /
        // (256,128,64)): the hypothesis that these GEMMs are bound by the ~12 B/clk a CU streams from beyond its L2 does not hold
        // for them -- four 64 x 64 workgroups per CU hide each other's latency better than one big one.  Nor are they bound by the MFMA
        // rate: the same GEMMs on the bf16 matrix pipe with split operands (as the PPO minibatch kernel, 2.7x the MFMA rate) measured
        // 206 / 254 / 290 us (profiles/r03_gemm_split_ab.txt): K is 64 .. 256, the pipeline fill and the staging dominate.
        static const int forced = [] { const char *e = getenv("ERL_GEMM_TILE"); return e ? atoi(e) : 0; }();
        int rm = 1, rn = 1;
        if (forced == 12 || forced == 21 || forced == 22) { rm = forced / 10; rn = forced % 10; }
        const dim3 grid((unsigned)erl_cdiv(g.N, GT * rn), (unsigned)erl_cdiv(g.M, GT * rm), (unsigned)splits);
        const size_t lds = (size_t)2 * GK * ((GT * rm + 4) + (GT * rn + 4)) * sizeof(float);
#define ERL_GEMM_GO(RM, RN)                                                                 \
    do {                                                                                    \
        if (va && vb) gemm_go<AOP, BOP, EPI, true, true, RM, RN>(grid, lds, s, g);           \
        else if (va) gemm_go<AOP, BOP, EPI, true, false, RM, RN>(grid, lds, s, g);           \
        else if (vb) gemm_go<AOP, BOP, EPI, false, true, RM, RN>(grid, lds, s, g);           \
        else gemm_go<AOP, BOP, EPI, false, false, RM, RN>(grid, lds, s, g);                  \
    } while (0)
        if (rm == 2 && rn == 2) ERL_GEMM_GO(2, 2);
        else if (rn == 2) ERL_GEMM_GO(1, 2);
        else if (rm == 2) ERL_GEMM_GO(2, 1);
        else ERL_GEMM_GO(1, 1);
#undef ERL_GEMM_GO
    }
    return erl_hip_status(hipGetLastError(), what);
}

// Y[rows][Nw] = act(X[rows][K] . W[Nw][K]^T + b);  gelu: hidden layer (G receives GELU' when not NULL)
int dense_forward(hipStream_t s, const float *X, const float *W, const float *b, float *Y, float *G, int rows, int Nw, int K, bool gelu)
{
    GemmArgs g{};
    g.A = X; g.lda = K; g.B = W; g.ldb = K; g.M = rows; g.N = Nw; g.K = K; g.C = Y; g.ldc = Nw; g.bias = b; g.G = G;
    return gelu ? gemm_launch<OP_RC, OP_RC, EPI_BIAS_GELU>(s, g, 1, "dense_forward") : gemm_launch<OP_RC, OP_RC, EPI_BIAS>(s, g, 1, "dense_forward");
}

// dX[rows][K] (=, +=, or = gate *) dZ[rows][Nw] . W[Nw][K]
int dense_backward_input(hipStream_t s, const float *dZ, const float *W, float *dX, const float *gate, bool accumulate, int rows, int Nw,
                         int K)
{
    GemmArgs g{};
    g.A = dZ; g.lda = Nw; g.B = W; g.ldb = K; g.M = rows; g.N = K; g.K = Nw; g.C = dX; g.ldc = K; g.G = const_cast<float *>(gate);
    if (gate) return gemm_launch<OP_RC, OP_OC, EPI_MUL>(s, g, 1, "dense_backward_input");
    return accumulate ? gemm_launch<OP_RC, OP_OC, EPI_ACC>(s, g, 1, "dense_backward_input")
                      : gemm_launch<OP_RC, OP_OC, EPI_STORE>(s, g, 1, "dense_backward_input");
}

}  // namespace
