// E5: stable LSD radix sort of (int64 key, int32 value) pairs — the depth sort within tiles
// (SURVEY.md §8a-6).  Replaces `cub::DeviceRadixSort::SortPairs` inside gsplat 1.0.0's
// isect_tiles (reached from /root/reference/dn_splatter/dn_model.py:570-591) and the
// `torch.sort` of the legacy path (dn_model.py:644-653).
//
// Written for wave64 and for a chip where a dependent kernel boundary costs as much as moving a
// megabyte: 8-bit digits, TWO launches per pass.  The input is cut into at most kMaxGroups
// contiguous chunks, one per workgroup (4 waves), each a whole number of 2048-key tiles:
//   hist     every workgroup counts the digits of its chunk -> hist[group][256]
//   scatter  every workgroup rebuilds its own bases from that small table (column sums over the
//            groups before it + an in-block scan over digits: no separate scan launches, no
//            inter-workgroup waiting), then walks its tiles in order, ranking keys and
//            scattering them.
// Inside a tile wave w owns the contiguous slice [w*512, (w+1)*512), so "earlier key" ==
// (tile, wave, round, lane) order and every load is a 512-byte coalesced row.  Lanes with equal
// digits find each other with 8 wave ballots (the 64-bit exec-mask form of a match-any), so
// ranking needs neither LDS atomics nor a per-lane counter array: one LDS counter row per wave,
// touched once per digit group.  Stability makes the result bit-identical to any other stable
// sort of the same keys.
#include "common.h"

namespace fsgs {

constexpr int kSortWaves = 4;
constexpr int kSortBlock = kSortWaves * 64;
constexpr int kSortRounds = 8;
constexpr int kSortTile = kSortBlock * kSortRounds;  // 2048 keys per tile
constexpr int kRadix = 256;
constexpr int kMaxGroups = 256;                      // one per CU

// Mask of lanes in this wave whose digit equals mine (inactive lanes pass active=false).
__device__ __forceinline__ unsigned long long match_digit(unsigned d, bool active) {
    unsigned long long m = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long vote = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? vote : ~vote;
    }
    return m;
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
    const int lane = threadIdx.x & 63;
    return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}

// hist[group][digit] over the group's chunk [group*chunk, min(n, (group+1)*chunk))
__global__ void __launch_bounds__(kSortBlock)
radix_hist_kernel(int64_t n, int64_t chunk, const uint64_t *__restrict__ keys, int shift,
                  int32_t *__restrict__ hist) {
    __shared__ unsigned counters[kSortWaves][kRadix];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < kSortWaves * kRadix; i += kSortBlock) (&counters[0][0])[i] = 0;
    __syncthreads();
    const int64_t begin = (int64_t)blockIdx.x * chunk;
    const int64_t end = min(n, begin + chunk);
    const unsigned long long lt = lanemask_lt();
    for (int64_t tile = begin; tile < end; tile += kSortTile) {
        const int64_t base = tile + (int64_t)w * (kSortRounds * 64);
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) {
            const int64_t i = base + r * 64 + lane;
            const bool act = i < end;
            const unsigned d = act ? (unsigned)((keys[i] >> shift) & 0xFFu) : 0u;
            const unsigned long long m = match_digit(d, act);
            if (act && (m & lt) == 0) counters[w][d] += (unsigned)__popcll(m);
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kRadix; d += kSortBlock) {
        unsigned s = 0;
#pragma unroll
        for (int k = 0; k < kSortWaves; ++k) s += counters[k][d];
        hist[(int64_t)blockIdx.x * kRadix + d] = (int32_t)s;
    }
}

__global__ void __launch_bounds__(kSortBlock)
radix_scatter_kernel(int64_t n, int64_t chunk, int n_groups, const uint64_t *__restrict__ keys_in,
                     const int32_t *__restrict__ vals_in, uint64_t *__restrict__ keys_out,
                     int32_t *__restrict__ vals_out, int shift, const int32_t *__restrict__ hist) {
    __shared__ unsigned counters[kSortWaves][kRadix];
    __shared__ int64_t bases[kSortWaves][kRadix];
    __shared__ int64_t running[kRadix];
    __shared__ int64_t wtot[kSortWaves];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x;

    // ---- this group's starting offset per digit, from the [group][digit] table ---------------
    {
        const int d = threadIdx.x;  // kSortBlock == kRadix
        int64_t col_total = 0, col_before = 0;
        for (int k = 0; k < n_groups; ++k) {
            const int v = hist[(int64_t)k * kRadix + d];
            if (k < g) col_before += v;
            col_total += v;
        }
        // exclusive scan of col_total over the 256 digits
        int64_t inc = col_total;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int64_t o = __shfl_up(inc, s, 64);
            if (lane >= s) inc += o;
        }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        int64_t wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wtot[k];
        running[d] = wbase + inc - col_total + col_before;
    }
    __syncthreads();

    const int64_t begin = (int64_t)g * chunk;
    const int64_t end = min(n, begin + chunk);
    const unsigned long long lt = lanemask_lt();
    for (int64_t tile = begin; tile < end; tile += kSortTile) {
        for (int i = threadIdx.x; i < kSortWaves * kRadix; i += kSortBlock) (&counters[0][0])[i] = 0;
        __syncthreads();
        const int64_t base = tile + (int64_t)w * (kSortRounds * 64);
        uint64_t key[kSortRounds];
        int32_t val[kSortRounds];
        unsigned rank[kSortRounds];
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) {
            const int64_t i = base + r * 64 + lane;
            const bool act = i < end;
            key[r] = act ? keys_in[i] : 0ull;
            val[r] = act ? vals_in[i] : 0;
        }
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) {
            const int64_t i = base + r * 64 + lane;
            const bool act = i < end;
            const unsigned d = (unsigned)((key[r] >> shift) & 0xFFu);
            const unsigned long long m = match_digit(d, act);
            const unsigned before = (unsigned)__popcll(m & lt);
            unsigned prev = 0;
            if (act) prev = counters[w][d];
            // every lane of the group has read `prev` (same wave, program order) before the
            // group leader bumps the counter
            if (act && before == 0) counters[w][d] = prev + (unsigned)__popcll(m);
            rank[r] = prev + before;
        }
        __syncthreads();
        {
            const int d = threadIdx.x;
            int64_t run = running[d];
#pragma unroll
            for (int k = 0; k < kSortWaves; ++k) {
                bases[k][d] = run;
                run += counters[k][d];
            }
            running[d] = run;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) {
            const int64_t i = base + r * 64 + lane;
            if (i < end) {
                const unsigned d = (unsigned)((key[r] >> shift) & 0xFFu);
                const int64_t pos = bases[w][d] + rank[r];
                keys_out[pos] = key[r];
                vals_out[pos] = val[r];
            }
        }
        __syncthreads();  // bases / counters are rewritten by the next tile
    }
}

struct SortPlan {
    int n_groups;
    int64_t chunk;
};

inline SortPlan sort_plan(int64_t n) {
    const int64_t tiles = (n + kSortTile - 1) / kSortTile;
    SortPlan p;
    p.n_groups = (int)(tiles < kMaxGroups ? (tiles > 0 ? tiles : 1) : kMaxGroups);
    const int64_t tiles_per_group = (tiles + p.n_groups - 1) / p.n_groups;
    p.chunk = (tiles_per_group > 0 ? tiles_per_group : 1) * kSortTile;
    // groups at the tail may be empty when tiles is not a multiple: shrink to the groups in use
    p.n_groups = (int)((n + p.chunk - 1) / p.chunk);
    if (p.n_groups < 1) p.n_groups = 1;
    return p;
}

}  // namespace fsgs

using namespace fsgs;

// scratch: hist int32 [n_groups * 256]
extern "C" size_t fsgs_sort_scratch_bytes(int64_t n) {
    (void)n;
    return (size_t)kMaxGroups * kRadix * sizeof(int32_t) + 64;
}

extern "C" int fsgs_sort_pairs(int64_t n, int64_t *keys_a, int32_t *vals_a, int64_t *keys_b,
                               int32_t *vals_b, int end_bit, void *scratch, size_t scratch_bytes,
                               int *result_in_b, fsgs_stream_t stream) {
    if (n < 0 || end_bit < 0 || end_bit > 64 || !result_in_b) return FSGS_EINVAL;
    *result_in_b = 0;
    if (n <= 1 || end_bit == 0) return FSGS_OK;
    if (!keys_a || !vals_a || !keys_b || !vals_b || !scratch) return FSGS_EINVAL;
    if (scratch_bytes < fsgs_sort_scratch_bytes(n)) return FSGS_ESCRATCH;
    const SortPlan plan = sort_plan(n);
    int32_t *hist = reinterpret_cast<int32_t *>(scratch);
    hipStream_t s = as_stream(stream);
    uint64_t *kin = reinterpret_cast<uint64_t *>(keys_a), *kout = reinterpret_cast<uint64_t *>(keys_b);
    int32_t *vin = vals_a, *vout = vals_b;
    int in_b = 0;
    for (int shift = 0; shift < end_bit; shift += 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(plan.n_groups), dim3(kSortBlock), 0, s, n, plan.chunk, kin,
                           shift, hist);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(plan.n_groups), dim3(kSortBlock), 0, s, n, plan.chunk,
                           plan.n_groups, kin, vin, kout, vout, shift, hist);
        uint64_t *tk = kin; kin = kout; kout = tk;
        int32_t *tv = vin; vin = vout; vout = tv;
        in_b ^= 1;
    }
    *result_in_b = in_b;
    return check_launch();
}
