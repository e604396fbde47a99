// E5: stable LSD radix sort of (int64 key, int32 value) pairs — the depth sort within tiles
// (SURVEY.md §8a-6).  Replaces `cub::DeviceRadixSort::SortPairs` inside gsplat 1.0.0's
// isect_tiles (reached from /root/reference/dn_splatter/dn_model.py:570-591) and the
// `torch.sort` of the legacy path (dn_model.py:644-653).
//
// Three launches per 8-bit pass (per-block digit histograms -> one workgroup per digit turns its row
// into a prefix over blocks -> scatter, which scans the 256 digit totals itself).
// Written for wave64: 8-bit digits; a workgroup of 4 waves owns a tile of 4 x R x 64 keys, wave w
// owning the contiguous slice [w*R*64, (w+1)*R*64) so that "earlier key" == (wave, round, lane)
// order and every load is a 512-byte coalesced row.  Lanes with equal digits find each other
// with 8 wave ballots (the 64-bit exec-mask form of a match-any), so ranking needs neither LDS
// atomics nor a per-lane counter array: one LDS counter row per wave, touched once per group.
// Stability makes the result bit-identical to any other stable sort of the same keys.
#include "common.h"

namespace fsgs {

constexpr int kSortWaves = 4;
constexpr int kSortBlock = kSortWaves * 64;
constexpr int kSortRounds = 8;
constexpr int kSortTile = kSortBlock * kSortRounds;  // 2048 keys per workgroup
constexpr int kRadix = 256;

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

// Per-workgroup digit histogram, written digit-major: hist[d * nblocks + block].
__global__ void __launch_bounds__(kSortBlock)
radix_hist_kernel(int64_t n, const uint64_t *__restrict__ keys, int shift, int nblocks,
                  int32_t *__restrict__ hist) {
    __shared__ unsigned counters[kSortWaves][kRadix];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < kSortWaves * kRadix; i += kSortBlock) (&counters[0][0])[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortTile + (int64_t)w * (kSortRounds * 64);
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = base + r * 64 + lane;
        const bool act = i < n;
        const unsigned d = act ? (unsigned)((keys[i] >> shift) & 0xFFu) : 0u;
        const unsigned long long m = match_digit(d, act);
        if (act && (m & lanemask_lt()) == 0) counters[w][d] += (unsigned)__popcll(m);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kRadix; d += kSortBlock) {
        unsigned s = 0;
#pragma unroll
        for (int k = 0; k < kSortWaves; ++k) s += counters[k][d];
        hist[(int64_t)d * nblocks + blockIdx.x] = (int32_t)s;
    }
}

// One workgroup per digit: exclusive prefix of hist[d][0..nblocks) over the blocks (in place, as
// int32: a digit's count never exceeds n < 2^31 on this path) and the digit's total.
__global__ void __launch_bounds__(kSortBlock)
radix_digit_scan_kernel(int nblocks, int32_t *__restrict__ hist, int64_t *__restrict__ digit_total) {
    __shared__ int wsum[kSortWaves];
    __shared__ int carry_s;
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    int32_t *row = hist + (int64_t)d * nblocks;
    for (int base = 0; base < nblocks; base += kSortBlock) {
        const int i = base + tid;
        const int v = (i < nblocks) ? row[i] : 0;
        int inc = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int o = __shfl_up(inc, s, 64);
            if (lane >= s) inc += o;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        const int carry = carry_s;
        if (i < nblocks) row[i] = carry + wbase + inc - v;
        __syncthreads();
        if (tid == kSortBlock - 1) carry_s = carry + wbase + inc;
        __syncthreads();
    }
    if (tid == 0) digit_total[d] = carry_s;
}

__global__ void __launch_bounds__(kSortBlock)
radix_scatter_kernel(int64_t n, const uint64_t *__restrict__ keys_in, const int32_t *__restrict__ vals_in,
                     uint64_t *__restrict__ keys_out, int32_t *__restrict__ vals_out, int shift,
                     int nblocks, const int32_t *__restrict__ hist_prefix,
                     const int64_t *__restrict__ digit_total) {
    __shared__ unsigned counters[kSortWaves][kRadix];
    __shared__ int64_t bases[kSortWaves][kRadix];
    __shared__ int64_t digit_base[kRadix];
    __shared__ int64_t wtot[kSortWaves];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    {   // exclusive scan of the 256 digit totals, redone by every workgroup (no extra launch)
        const int64_t tot = digit_total[threadIdx.x];
        int64_t inc = tot;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int64_t o = __shfl_up(inc, s, 64);
            if (lane >= s) inc += o;
        }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        int64_t wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wtot[k];
        digit_base[threadIdx.x] = wbase + inc - tot;
    }
    for (int i = threadIdx.x; i < kSortWaves * kRadix; i += kSortBlock) (&counters[0][0])[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortTile + (int64_t)w * (kSortRounds * 64);
    uint64_t key[kSortRounds];
    int32_t val[kSortRounds];
    unsigned rank[kSortRounds];
    const unsigned long long lt = lanemask_lt();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = base + r * 64 + lane;
        const bool act = i < n;
        key[r] = act ? keys_in[i] : 0ull;
        val[r] = act ? vals_in[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = base + r * 64 + lane;
        const bool act = i < n;
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
    for (int d = threadIdx.x; d < kRadix; d += kSortBlock) {
        int64_t run = digit_base[d] + hist_prefix[(int64_t)d * nblocks + blockIdx.x];
#pragma unroll
        for (int k = 0; k < kSortWaves; ++k) {
            bases[k][d] = run;
            run += counters[k][d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = base + r * 64 + lane;
        if (i < n) {
            const unsigned d = (unsigned)((key[r] >> shift) & 0xFFu);
            const int64_t pos = bases[w][d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = val[r];
        }
    }
}

inline int64_t sort_num_blocks(int64_t n) { return (n + kSortTile - 1) / kSortTile; }

}  // namespace fsgs

using namespace fsgs;

// scratch layout: [hist int32: 256*nb (turned into per-digit block prefixes in place)] [digit_total i64: 256]
extern "C" size_t fsgs_sort_scratch_bytes(int64_t n) {
    const int64_t nb = sort_num_blocks(n > 0 ? n : 1);
    size_t bytes = (size_t)kRadix * nb * sizeof(int32_t);
    bytes = (bytes + 15) & ~(size_t)15;
    return bytes + kRadix * sizeof(int64_t) + 64;
}

extern "C" int fsgs_sort_pairs(int64_t n, int64_t *keys_a, int32_t *vals_a, int64_t *keys_b,
                               int32_t *vals_b, int end_bit, void *scratch, size_t scratch_bytes,
                               int *result_in_b, fsgs_stream_t stream) {
    if (n < 0 || end_bit < 0 || end_bit > 64 || !result_in_b) return FSGS_EINVAL;
    *result_in_b = 0;
    if (n <= 1 || end_bit == 0) return FSGS_OK;
    if (n >= (1ll << 31)) return FSGS_EINVAL;  // 32-bit list indices, as in the reference
    if (!keys_a || !vals_a || !keys_b || !vals_b || !scratch) return FSGS_EINVAL;
    if (scratch_bytes < fsgs_sort_scratch_bytes(n)) return FSGS_ESCRATCH;
    const int64_t nb = sort_num_blocks(n);
    char *p = reinterpret_cast<char *>(scratch);
    int32_t *hist = reinterpret_cast<int32_t *>(p);
    const size_t off = ((size_t)kRadix * nb * sizeof(int32_t) + 15) & ~(size_t)15;
    int64_t *digit_total = reinterpret_cast<int64_t *>(p + off);
    hipStream_t s = as_stream(stream);
    uint64_t *kin = reinterpret_cast<uint64_t *>(keys_a), *kout = reinterpret_cast<uint64_t *>(keys_b);
    int32_t *vin = vals_a, *vout = vals_b;
    int in_b = 0;
    // three launches per 8-bit pass: per-block histograms, per-digit prefix over blocks, scatter
    for (int shift = 0; shift < end_bit; shift += 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)nb), dim3(kSortBlock), 0, s, n, kin, shift,
                           (int)nb, hist);
        hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(kRadix), dim3(kSortBlock), 0, s, (int)nb, hist,
                           digit_total);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)nb), dim3(kSortBlock), 0, s, n, kin, vin,
                           kout, vout, shift, (int)nb, hist, digit_total);
        uint64_t *tk = kin; kin = kout; kout = tk;
        int32_t *tv = vin; vin = vout; vout = tv;
        in_b ^= 1;
    }
    *result_in_b = in_b;
    return check_launch();
}


