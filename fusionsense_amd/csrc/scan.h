// Device-wide prefix sums used by tile binning, the radix sort and densify compaction.
// Two launches (block reduce -> block scan + offset, where every workgroup sums the totals of the
// workgroups before it by itself): 2 reads + 1 write per element, no inter-workgroup spinning, so it
// is placement-independent, and one dependent launch fewer than a separate scan of the totals.
#pragma once
#include "common.h"

namespace fsgs {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;                          // items per thread
constexpr int kScanTile = kScanBlock * kScanItems;     // 2048 items per workgroup

inline int64_t scan_num_blocks(int64_t n) { return (n + kScanTile - 1) / kScanTile; }

// wave64 inclusive scan (Hillis-Steele on DPP-free shuffles; called a handful of times per block)
__device__ __forceinline__ int64_t wave_inclusive_scan(int64_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Block-wide exclusive scan of one value per thread (256 threads = 4 waves). Returns the
// exclusive prefix; *total receives the block sum.
__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t *total, int64_t *lds4) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t inc = wave_inclusive_scan(v);
    if (lane == 63) lds4[w] = inc;
    __syncthreads();
    int64_t base = 0;
#pragma unroll
    for (int i = 0; i < kScanBlock / 64; ++i) {
        if (i < w) base += lds4[i];
    }
    int64_t tot = 0;
#pragma unroll
    for (int i = 0; i < kScanBlock / 64; ++i) tot += lds4[i];
    *total = tot;
    __syncthreads();
    return base + inc - v;
}

template <typename InT>
__global__ void __launch_bounds__(kScanBlock)
scan_reduce_kernel(int64_t n, const InT *__restrict__ in, int64_t *__restrict__ block_sums) {
    __shared__ int64_t lds4[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k * kScanBlock + threadIdx.x;
        if (i < n) s += (int64_t)in[i];
    }
    int64_t tot;
    block_exclusive_scan(s, &tot, lds4);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// out[i] = (INCLUSIVE ? in[i] : 0) + sum_{j<i} in[j]
template <typename InT, bool INCLUSIVE>
__global__ void __launch_bounds__(kScanBlock)
scan_apply_kernel(int64_t n, const InT *__restrict__ in, const int64_t *__restrict__ block_sums,
                  int64_t *__restrict__ out) {
    __shared__ int64_t lds4[4];
    // blocked arrangement: thread t owns items [t*kScanItems, (t+1)*kScanItems) of the tile
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int64_t v[kScanItems];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        v[k] = (i < n) ? (int64_t)in[i] : 0;
        s += v[k];
    }
    // prefix of this workgroup = sum of the totals of all earlier workgroups
    int64_t before = 0;
    for (int64_t b = threadIdx.x; b < (int64_t)blockIdx.x; b += kScanBlock) before += block_sums[b];
    int64_t btot;
    block_exclusive_scan(before, &btot, lds4);
    int64_t tot;
    int64_t run = block_exclusive_scan(s, &tot, lds4) + btot;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        if (i < n) out[i] = INCLUSIVE ? run + v[k] : run;
        run += v[k];
    }
}

template <typename InT, bool INCLUSIVE>
inline int device_scan(int64_t n, const InT *in, int64_t *out, void *scratch, size_t scratch_bytes,
                       hipStream_t stream) {
    if (n == 0) return FSGS_OK;
    const int64_t nb = scan_num_blocks(n);
    if (scratch_bytes < (size_t)nb * sizeof(int64_t) || !scratch) return FSGS_ESCRATCH;
    int64_t *block_sums = reinterpret_cast<int64_t *>(scratch);
    hipLaunchKernelGGL((scan_reduce_kernel<InT>), dim3((unsigned)nb), dim3(kScanBlock), 0, stream, n, in,
                       block_sums);
    hipLaunchKernelGGL((scan_apply_kernel<InT, INCLUSIVE>), dim3((unsigned)nb), dim3(kScanBlock), 0,
                       stream, n, in, block_sums, out);
    return check_launch();
}

}  // namespace fsgs
