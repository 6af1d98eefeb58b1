// Prioritised experience replay on the device: one sum tree + one min tree per sequence as implicit
// heaps in HBM, replacing the reference's per-sequence CPU SumTree with its Python loop per sample and .cpu() round trips
// (elegantrl/train/replay_buffer.py:136-179, :226-299).  gfx950.
//
// Layout: sum / mn are (Q, 2 L) fp32, L = next power of two >= max_size, node 1 = root, leaf of time row r at L + r; unwritten
// leaves are 0 in the sum tree and +inf in the min tree.  Parents are always RECOMPUTED as left + right (never incremented), level
// by level, so the trees are a deterministic function of the leaf values; duplicate (row, sequence) pairs of one update list are
// resolved deterministically (highest list index wins, as numpy's assignment in the oracle) -- bit-exact against
// oracle/per_numpy.py, which also lists how this corrected restatement deviates from the reference's (non-working) SumTree.
#include "erl_common.h"

namespace {

constexpr int PER_T = 1024;

// items -> (leaf row, sequence, priority)
struct PerItems {
    // explicit list (td_error_update_for_per): ids0 / ids1 int64, td_error
    const int64_t *ids0, *ids1;
    const float *td;
    float alpha;
    // or a row range of every sequence (ReplayBuffer.update): rows (start + i / Q) mod max_size, priority `prob`
    int64_t start, max_size;
    float prob;
    int64_t n;
    int Q;
};

// returns false for an explicit item whose (row, sequence) lies outside the trees: it is skipped (the ids are device data the
// host cannot validate without a sync)
__device__ __forceinline__ bool per_item(const PerItems &it, int64_t i, int64_t &row, int &q, float &p)
{
    if (it.ids0) {
        row = it.ids0[i];
        const int64_t q64 = it.ids1[i];
        q = (int)q64;
        const float t = fminf(fmaxf(it.td[i], 1e-8f), 10.f);            // td_error.clamp(1e-8, 10).pow(per_alpha)   (:168)
        p = it.alpha == 1.f ? t : powf(t, it.alpha);                    // (alpha = 1: the priorities as given, exactly)
        return row >= 0 && row < it.max_size && q64 >= 0 && q64 < it.Q;
    } else {
        const int64_t r = i / it.Q;
        q = (int)(i - r * it.Q);
        row = it.start + r;
        if (row >= it.max_size) row -= it.max_size;
        p = it.prob;
        return true;
    }
}

// ONE workgroup: leaves, then one level per barrier; threads that share a parent write the same value (left + right).
// Duplicate (row, sequence) pairs in one explicit list (stratified draws / the successor clamp can repeat a transition) are
// resolved like the oracle's numpy assignment: the HIGHEST list index wins -- the items claim their leaf with an atomic max
// on the list index (the min tree's leaf is the claim slot), and only the winner writes the priority.
__global__ __launch_bounds__(PER_T) void per_update_kernel(float *__restrict__ sum, float *__restrict__ mn, int64_t L, PerItems it)
{
    const int64_t twoL = 2 * L;
    unsigned win = 0xffffffffu;                            // bit k: this thread's k-th item writes its leaf (n <= 8 * PER_T here)
    if (it.ids0) {
        for (int64_t i = threadIdx.x; i < it.n; i += PER_T) {
            int64_t row; int q; float p;
            if (per_item(it, i, row, q, p)) reinterpret_cast<int *>(mn)[(int64_t)q * twoL + L + row] = -1;
        }
        __syncthreads();
        for (int64_t i = threadIdx.x; i < it.n; i += PER_T) {
            int64_t row; int q; float p;
            if (per_item(it, i, row, q, p)) atomicMax(reinterpret_cast<int *>(mn) + (int64_t)q * twoL + L + row, (int)i);
        }
        __syncthreads();
        win = 0u;
        int k = 0;
        for (int64_t i = threadIdx.x; i < it.n; i += PER_T, ++k) {
            int64_t row; int q; float p;
            if (per_item(it, i, row, q, p) && reinterpret_cast<const int *>(mn)[(int64_t)q * twoL + L + row] == (int)i) win |= 1u << k;
        }
        __syncthreads();
    }
    {
        int k = 0;
        for (int64_t i = threadIdx.x; i < it.n; i += PER_T, ++k) {
            int64_t row; int q; float p;
            if (per_item(it, i, row, q, p) && ((win >> k) & 1u)) {
                sum[(int64_t)q * twoL + L + row] = p;
                mn[(int64_t)q * twoL + L + row] = p;
            }
        }
    }
    for (int sh = 1; (L >> (sh - 1)) > 1; ++sh) {          // parents of the nodes (L + row) >> (sh - 1): one tree level per barrier
        __syncthreads();                                   // (waits for this wave's stores: the level below is complete)
        for (int64_t i = threadIdx.x; i < it.n; i += PER_T) {
            int64_t row; int q; float p;
            if (!per_item(it, i, row, q, p)) continue;
            const int64_t par = (L + row) >> sh, base = (int64_t)q * twoL;
            sum[base + par] = sum[base + 2 * par] + sum[base + 2 * par + 1];
            mn[base + par] = fminf(mn[base + 2 * par], mn[base + 2 * par + 1]);
        }
    }
}

// bulk path: leaves in parallel, then every internal level in its own launch.  `phase` (explicit lists only, see
// per_update_kernel): 0 reset the claim slots, 1 claim, 2 the winners write the sum leaf, 3 every item copies it to the min leaf;
// row ranges (no duplicates) write both leaves in one pass (phase -1).
__global__ __launch_bounds__(256) void per_leaves_kernel(float *__restrict__ sum, float *__restrict__ mn, int64_t L, PerItems it, int phase)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= it.n) return;
    int64_t row; int q; float p;
    if (!per_item(it, i, row, q, p)) return;
    const int64_t leaf = (int64_t)q * 2 * L + L + row;
    switch (phase) {
        case 0: reinterpret_cast<int *>(mn)[leaf] = -1; break;
        case 1: atomicMax(reinterpret_cast<int *>(mn) + leaf, (int)i); break;
        case 2: if (reinterpret_cast<const int *>(mn)[leaf] == (int)i) sum[leaf] = p; break;
        case 3: mn[leaf] = sum[leaf]; break;
        default: sum[leaf] = p; mn[leaf] = p; break;
    }
}

__global__ __launch_bounds__(256) void per_level_kernel(float *__restrict__ sum, float *__restrict__ mn, int64_t L, int64_t half, int Q)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // over Q * half / 2 parents of the level [half, 2 half)
    const int64_t per = half >> 1;
    if (i >= (int64_t)Q * per) return;
    const int64_t q = i / per, par = per + (i - q * per), base = q * 2 * L;
    sum[base + par] = sum[base + 2 * par] + sum[base + 2 * par + 1];
    mn[base + par] = fminf(mn[base + 2 * par], mn[base + 2 * par + 1]);
}

__global__ __launch_bounds__(256) void per_init_kernel(float *__restrict__ sum, float *__restrict__ mn, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        sum[i] = 0.f;
        mn[i] = __builtin_inff();
    }
}

// proportional prioritisation with stratified draws (:285-298), full-depth descent; one thread per sample
__global__ __launch_bounds__(256) void per_sample_kernel(const float *__restrict__ sum, const float *__restrict__ mn, int64_t L, int Q,
                                                         const float *__restrict__ uniform, int64_t n, int64_t cur_size, int64_t newest,
                                                         float beta, int64_t *__restrict__ out_index, float *__restrict__ out_weight)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Q * n) return;
    const int64_t q = i / n, j = i - q * n, base = q * 2 * L;
    const float total = sum[base + 1];
    float v = ((float)j + uniform[i]) * (total / (float)n);             // (arange(n) + rand(n)) * (tree[0] / n)   (:287)
    int64_t node = 1;
    while (node < L) {
        const float left = sum[base + 2 * node];
        if (v <= left) node = 2 * node;
        else {
            v -= left;
            node = 2 * node + 1;
        }
    }
    int64_t row = node - L;
    if (row > cur_size - 2) row = cur_size - 2;                        // the last filled position has no successor row (oracle D4)
    if (row == newest) row = newest >= 1 ? newest - 1 : 1;             // full ring: the newest row's successor slot holds the OLDEST data (D6)
    out_index[i] = q * cur_size + row;                                  // decodes by the reference's fmod / div (:155-156)
    out_weight[i] = powf(sum[base + L + row] / mn[base + 1], -beta);    // (prob / min prob)^(-beta)   (:296-297)
}

int64_t per_leaves(int64_t max_size)
{
    int64_t L = 2;
    while (L < max_size) L <<= 1;
    return L;
}

int per_update_impl(const char *what, float *sum, float *mn, int64_t max_size, int Q, PerItems it, hipStream_t s)
{
    const int64_t L = per_leaves(max_size);
    if (it.n == 0) return ERL_OK;
    if (it.n <= 8 * PER_T) {
        hipLaunchKernelGGL(per_update_kernel, dim3(1), dim3(PER_T), 0, s, sum, mn, L, it);
    } else {
        const dim3 grid((unsigned)erl_cdiv(it.n, 256));
        if (it.ids0)
            for (int phase = 0; phase < 4; ++phase) hipLaunchKernelGGL(per_leaves_kernel, grid, dim3(256), 0, s, sum, mn, L, it, phase);
        else
            hipLaunchKernelGGL(per_leaves_kernel, grid, dim3(256), 0, s, sum, mn, L, it, -1);
        for (int64_t half = L; half > 1; half >>= 1)
            hipLaunchKernelGGL(per_level_kernel, dim3((unsigned)erl_cdiv((int64_t)Q * (half >> 1), 256)), dim3(256), 0, s, sum, mn, L,
                               half, Q);
    }
    return erl_hip_status(hipGetLastError(), what);
}

}  // namespace

extern "C" int64_t erl_per_tree_floats(int64_t max_size, int64_t num_seqs)
{
    if (max_size < 2 || num_seqs < 1 || max_size > (1LL << 30)) return -1;
    return num_seqs * 2 * per_leaves(max_size);
}

extern "C" int erl_per_init_f32(float *sum_tree, float *min_tree, int64_t max_size, int64_t num_seqs, void *stream)
{
    ERL_REQUIRE(sum_tree && min_tree, "erl_per_init_f32: NULL tensor");
    const int64_t n = erl_per_tree_floats(max_size, num_seqs);
    ERL_REQUIRE(n > 0, "erl_per_init_f32: bad shape");
    hipLaunchKernelGGL(per_init_kernel, dim3((unsigned)erl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, sum_tree, min_tree, n);
    ERL_LAUNCH_CHECK("erl_per_init_f32");
}

extern "C" int erl_per_add_rows_f32(float *sum_tree, float *min_tree, int64_t max_size, int64_t num_seqs, int64_t start, int64_t add,
                                    float prob, void *stream)
{
    ERL_REQUIRE(sum_tree && min_tree, "erl_per_add_rows_f32: NULL tensor");
    ERL_REQUIRE(max_size >= 2 && num_seqs >= 1 && num_seqs < (1 << 30) && start >= 0 && start <= max_size && add >= 0 && add <= max_size,
                "erl_per_add_rows_f32: bad argument");
    PerItems it{};
    it.start = start == max_size ? 0 : start;
    it.max_size = max_size;
    it.prob = prob;
    it.n = add * num_seqs;
    it.Q = (int)num_seqs;
    return per_update_impl("erl_per_add_rows_f32", sum_tree, min_tree, max_size, (int)num_seqs, it, (hipStream_t)stream);
}

extern "C" int erl_per_update_f32(float *sum_tree, float *min_tree, int64_t max_size, int64_t num_seqs, const int64_t *ids0,
                                  const int64_t *ids1, const float *td_error, int64_t n, float per_alpha, void *stream)
{
    ERL_REQUIRE(sum_tree && min_tree && ids0 && ids1 && td_error, "erl_per_update_f32: NULL tensor");
    ERL_REQUIRE(max_size >= 2 && num_seqs >= 1 && num_seqs < (1 << 30) && n >= 0 && n < (1LL << 31), "erl_per_update_f32: bad argument");
    PerItems it{};
    it.ids0 = ids0; it.ids1 = ids1; it.td = td_error; it.alpha = per_alpha;
    it.max_size = max_size;
    it.n = n;
    it.Q = (int)num_seqs;
    return per_update_impl("erl_per_update_f32", sum_tree, min_tree, max_size, (int)num_seqs, it, (hipStream_t)stream);
}

extern "C" int erl_per_sample_f32(const float *sum_tree, const float *min_tree, int64_t max_size, int64_t num_seqs,
                                  const float *uniform, int64_t n_per_seq, int64_t cur_size, int64_t cursor, float per_beta,
                                  int64_t *out_index, float *out_weight, void *stream)
{
    ERL_REQUIRE(sum_tree && min_tree && uniform && out_index && out_weight, "erl_per_sample_f32: NULL tensor");
    ERL_REQUIRE(max_size >= 2 && num_seqs >= 1 && n_per_seq >= 1 && cur_size >= 2 && cur_size <= max_size && cursor <= max_size,
                "erl_per_sample_f32: bad argument (cur_size=%lld max_size=%lld cursor=%lld)", (long long)cur_size, (long long)max_size,
                (long long)cursor);
    const int64_t L = per_leaves(max_size);
    // the ring's write cursor `p` when the ring is full (cursor < 0: not full / unknown): the newest row (p - 1) is followed in
    // memory by the oldest one, so it has no valid successor either
    const int64_t newest = (cursor >= 0 && cur_size == max_size && max_size >= 3) ? (cursor + max_size - 1) % max_size : -1;
    hipLaunchKernelGGL(per_sample_kernel, dim3((unsigned)erl_cdiv(num_seqs * n_per_seq, 256)), dim3(256), 0, (hipStream_t)stream, sum_tree,
                       min_tree, L, (int)num_seqs, uniform, n_per_seq, cur_size, newest, per_beta, out_index, out_weight);
    ERL_LAUNCH_CHECK("erl_per_sample_f32");
}
