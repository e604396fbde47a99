// E4-E6 in four launches: tile-bucketed binning + per-tile LDS sort.
// Replaces gsplat 1.0.0's isect_tiles + cub::DeviceRadixSort::SortPairs + isect_offset_encode
// (reached from /root/reference/dn_splatter/dn_model.py:570-591) with the same outputs, bit for bit.
//
// The global 44-bit LSD radix sort of (tile | depth) keys is pass- and launch-bound on this chip
// (6 passes x 5 dependent launches of ~5-15 us each for M = 2 M keys).  But the high key bits are
// the tile index, which every Gaussian knows when it emits its intersections.  So:
//   A1  per Gaussian: tile rectangle, tiles_per_gauss, and one atomic increment per touched tile
//   A2  one workgroup: exclusive scan of the per-tile counts = isect_offsets (no offset_encode
//       pass), total M and the largest bucket; one host sync (the same one the reference has)
//   B1  per Gaussian: claim a slot in each touched tile's bucket (returning atomic on the tile's
//       cursor) and drop the 64-bit in-tile key  bits(depth) << 32 | flatten_id  there
//   B2  one workgroup per tile: bitonic sort of the bucket in LDS (160 KiB per CU: buckets up to
//       8192 entries), then write isect_ids = cam|tile|depth and flatten_ids in sorted order.
// Slot claiming order is non-deterministic, the per-tile sort on (depth, id) makes the result
// deterministic and identical to a stable sort of the emission order (ties -> ascending id).
#include "common.h"

namespace fsgs {

struct TileRectB {
    int x0, y0, x1, y1;
};

__device__ __forceinline__ TileRectB tile_rect_b(float mx, float my, int radius, int tile_size, int tw, int th,
                                                 int legacy) {
    const float ts = (float)tile_size;
    const float tr = (float)radius / ts;
    const float tx = mx / ts, ty = my / ts;
    float lx = tx - tr, ly = ty - tr, hx = tx + tr, hy = ty + tr;
    if (legacy) {
        lx = truncf(lx); ly = truncf(ly);
        hx = truncf(hx + 1.f); hy = truncf(hy + 1.f);
    } else {
        lx = floorf(lx); ly = floorf(ly);
        hx = ceilf(hx); hy = ceilf(hy);
    }
    TileRectB r;
    r.x0 = (int)fminf(fmaxf(lx, 0.f), (float)tw);
    r.y0 = (int)fminf(fmaxf(ly, 0.f), (float)th);
    r.x1 = (int)fminf(fmaxf(hx, 0.f), (float)tw);
    r.y1 = (int)fminf(fmaxf(hy, 0.f), (float)th);
    return r;
}

constexpr int kWideB = 64;  // Gaussians touching >= this many tiles are spread over their wave
// ~800 increments land on every tile counter; same-address atomics serialise at the memory side, so
// each tile gets kShards counters (one per wave index mod kShards), stored shard-major so that a
// tile's shards sit in different cache lines.  Bucket (tile, shard) is the contiguous range
// [offsets[tile*kShards+shard], offsets[tile*kShards+shard+1]); a tile's shards are adjacent.
constexpr int kShards = 16;
__device__ __forceinline__ int shard_of_wave() { return (int)((blockIdx.x * 4 + (threadIdx.x >> 6)) & (kShards - 1)); }

// FILL = false: count pass (A1).  FILL = true: scatter pass (B1).
template <bool FILL>
__global__ void __launch_bounds__(256)
bucket_pass_kernel(int C, int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                   const float *__restrict__ depths, int tile_size, int tw, int th, int legacy,
                   int32_t *__restrict__ tiles_per_gauss, int32_t *__restrict__ tile_counts,
                   const int32_t *__restrict__ shard_offsets, unsigned long long *__restrict__ keys,
                   unsigned long long *__restrict__ rule_diff) {
    const int64_t total = (int64_t)C * N;
    const int T = C * tw * th;
    int32_t *my_counts = tile_counts + (int64_t)shard_of_wave() * T;   // shard-major counters
    const int sh = shard_of_wave();
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int n_tiles = tw * th;
    TileRectB t = {0, 0, 0, 0};
    int cnt = 0, tile_base = 0;
    unsigned long long key = 0ull;
    if (idx < total) {
        const int r = radii[idx];
        if (r > 0) {
            const float2 m = reinterpret_cast<const float2 *>(means2d)[idx];
            t = tile_rect_b(m.x, m.y, r, tile_size, tw, th, legacy);
            cnt = (t.x1 - t.x0) * (t.y1 - t.y0);
            tile_base = (int)(idx / N) * n_tiles;
            if (FILL) {
                key = ((unsigned long long)(uint32_t)__float_as_int(depths[idx]) << 32) | (uint32_t)idx;
            } else if (rule_diff) {
                const TileRectB o = tile_rect_b(m.x, m.y, r, tile_size, tw, th, !legacy);
                if (o.x0 != t.x0 || o.y0 != t.y0 || o.x1 != t.x1 || o.y1 != t.y1) atomicAdd(rule_diff, 1ull);
            }
        }
        if (!FILL) tiles_per_gauss[idx] = cnt;
    }
    const bool wide = cnt >= kWideB;
    if (cnt > 0 && !wide) {
        for (int y = t.y0; y < t.y1; ++y)
            for (int x = t.x0; x < t.x1; ++x) {
                const int tile = tile_base + y * tw + x;
                if (FILL) {
                    const int slot = atomicAdd(&my_counts[tile], 1);
                    keys[(int64_t)shard_offsets[tile * kShards + sh] + slot] = key;
                } else {
                    atomicAdd(&my_counts[tile], 1);
                }
            }
    }
    unsigned long long wide_mask = __ballot(wide);
    while (wide_mask) {
        const int src = __ffsll((long long)wide_mask) - 1;
        wide_mask &= wide_mask - 1;
        const int x0 = __shfl(t.x0, src, 64), y0 = __shfl(t.y0, src, 64), x1 = __shfl(t.x1, src, 64);
        const int n = __shfl(cnt, src, 64), tb = __shfl(tile_base, src, 64);
        const unsigned long long k = __shfl(key, src, 64);
        const int wdt = x1 - x0;
        for (int j = lane; j < n; j += 64) {
            const int tile = tb + (y0 + j / wdt) * tw + x0 + j % wdt;
            if (FILL) {
                const int slot = atomicAdd(&my_counts[tile], 1);
                keys[(int64_t)shard_offsets[tile * kShards + sh] + slot] = k;
            } else {
                atomicAdd(&my_counts[tile], 1);
            }
        }
    }
}

// A2: exclusive scan, by one workgroup, of the T*kShards counters visited tile-major (the counters are
// stored shard-major): shard_offsets[T*kShards + 1], tile_offsets[T + 1] (every kShards-th value),
// stats[0] = total, stats[1] = largest TILE bucket.  Re-zeroes the counters: they become B1's cursors.
__global__ void __launch_bounds__(1024)
tile_scan_kernel(int T, int32_t *__restrict__ tile_counts, int32_t *__restrict__ shard_offsets,
                 int32_t *__restrict__ tile_offsets, long long *__restrict__ stats) {
    __shared__ int wsum[16];
    __shared__ int carry_s, max_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { carry_s = 0; max_s = 0; }
    __syncthreads();
    const int n = T * kShards;
    int local_max = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        int v = 0;
        if (i < n) {
            const int tile = i / kShards, sh = i - tile * kShards;
            const int64_t src = (int64_t)sh * T + tile;
            v = tile_counts[src];
            tile_counts[src] = 0;
        }
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        // size of each tile's bucket = sum over its kShards consecutive entries (kShards | 64)
        int tsum = v;
#pragma unroll
        for (int d = 1; d < kShards; d <<= 1) tsum += __shfl_xor(tsum, d, 64);
        local_max = max(local_max, tsum);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        const int carry = carry_s;
        if (i < n) {
            const int ex = carry + wbase + inc - v;
            shard_offsets[i] = ex;
            if ((i & (kShards - 1)) == 0) tile_offsets[i / kShards] = ex;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wbase + inc;
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local_max = max(local_max, __shfl_xor(local_max, d, 64));
    if (lane == 0) atomicMax(&max_s, local_max);
    __syncthreads();
    if (tid == 0) {
        shard_offsets[n] = carry_s;
        tile_offsets[T] = carry_s;
        stats[0] = carry_s;
        stats[1] = max_s;
    }
}

// B2: one workgroup per (camera, tile): bitonic sort of the bucket's 64-bit keys in LDS.
__global__ void __launch_bounds__(256)
tile_sort_kernel(int n_tiles, int tile_bits, const int32_t *__restrict__ tile_offsets,
                 const unsigned long long *__restrict__ keys, int64_t *__restrict__ isect_ids,
                 int32_t *__restrict__ flatten_ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
    const int tile_lin = blockIdx.x;
    const int s = tile_offsets[tile_lin], e = tile_offsets[tile_lin + 1];
    const int n = e - s;
    if (n <= 0) return;
    int np = 1;
    while (np < n) np <<= 1;
    const int tid = threadIdx.x;
    for (int i = tid; i < np; i += 256) sk[i] = (i < n) ? keys[(int64_t)s + i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (np >> 1); i += 256) {
                // a = i with a zero bit inserted at position log2(j)
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int b = a | j;
                const bool up = (a & k) == 0;
                const unsigned long long ka = sk[a], kb = sk[b];
                if ((ka > kb) == up) { sk[a] = kb; sk[b] = ka; }
            }
            __syncthreads();
        }
    }
    const int cam = tile_lin / n_tiles, tile = tile_lin - cam * n_tiles;
    const int64_t hi = ((int64_t)cam << (32 + tile_bits)) | ((int64_t)tile << 32);
    for (int i = tid; i < n; i += 256) {
        const unsigned long long k = sk[i];
        isect_ids[(int64_t)s + i] = hi | (int64_t)(k >> 32);
        flatten_ids[(int64_t)s + i] = (int32_t)(k & 0xFFFFFFFFull);
    }
}

constexpr int kMaxBucket = 8192;  // 64 KiB of LDS keys per workgroup

}  // namespace fsgs

using namespace fsgs;

extern "C" int fsgs_bucket_max_tile(void) { return kMaxBucket; }
extern "C" int fsgs_bucket_shards(void) { return kShards; }

// tile_counts[T*fsgs_bucket_shards()] (scratch, becomes the zeroed cursor array), shard_offsets[T*shards+1],
// tile_offsets[T+1] (= isect_offsets + total), stats_dev[3] i64 scratch.
// Synchronises; host outputs: n_isects, largest bucket, rule_diff (nullable).
extern "C" int fsgs_bucket_count(int C, int N, const float *means2d, const int32_t *radii, int tile_size,
                                 int tile_width, int tile_height, int legacy, int32_t *tiles_per_gauss,
                                 int32_t *tile_counts, int32_t *shard_offsets, int32_t *tile_offsets,
                                 int64_t *stats_dev,
                                 int64_t *n_isects_host, int64_t *max_bucket_host, int64_t *n_rule_diff_host,
                                 fsgs_stream_t stream) {
    if (C < 0 || N < 0 || tile_size < 1 || !n_isects_host || !max_bucket_host) return FSGS_EINVAL;
    const int64_t total = (int64_t)C * N;
    const int T = C * tile_width * tile_height;
    *n_isects_host = 0;
    *max_bucket_host = 0;
    if (n_rule_diff_host) *n_rule_diff_host = 0;
    if (!tile_counts || !shard_offsets || !tile_offsets || !stats_dev) return FSGS_EINVAL;
    if (total > 0 && (!means2d || !radii || !tiles_per_gauss)) return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
    hipError_t e = hipMemsetAsync(tile_counts, 0, (size_t)T * kShards * sizeof(int32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(stats_dev, 0, 3 * sizeof(int64_t), s);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
    unsigned long long *diff_dev = n_rule_diff_host ? reinterpret_cast<unsigned long long *>(stats_dev + 2) : nullptr;
    if (total > 0)
        hipLaunchKernelGGL((bucket_pass_kernel<false>), dim3(ceil_div(total, 256)), dim3(256), 0, s, C, N, means2d,
                           radii, nullptr, tile_size, tile_width, tile_height, legacy, tiles_per_gauss,
                           tile_counts, nullptr, nullptr, diff_dev);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, T, tile_counts, shard_offsets, tile_offsets,
                       reinterpret_cast<long long *>(stats_dev));
    int64_t host[3] = {0, 0, 0};
    e = hipMemcpyAsync(host, stats_dev, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
    *n_isects_host = host[0];
    *max_bucket_host = host[1];
    if (n_rule_diff_host) *n_rule_diff_host = host[2];
    return check_launch();
}

// keys_scratch[M] u64.  max_bucket must be <= fsgs_bucket_max_tile().
extern "C" int fsgs_bucket_fill_sort(int C, int N, const float *means2d, const int32_t *radii,
                                     const float *depths, int tile_size, int tile_width, int tile_height,
                                     int legacy, int32_t *tile_cursors, const int32_t *shard_offsets,
                                     const int32_t *tile_offsets, int64_t n_isects, int64_t max_bucket, void *keys_scratch,
                                     int64_t *isect_ids, int32_t *flatten_ids, fsgs_stream_t stream) {
    if (C < 0 || N < 0 || tile_size < 1 || n_isects < 0 || max_bucket < 0 || max_bucket > kMaxBucket)
        return FSGS_EINVAL;
    if (n_isects == 0) return FSGS_OK;
    if (!means2d || !radii || !depths || !tile_cursors || !shard_offsets || !tile_offsets || !keys_scratch || !isect_ids ||
        !flatten_ids)
        return FSGS_EINVAL;
    const int64_t total = (int64_t)C * N;
    const int n_tiles = tile_width * tile_height;
    const int T = C * n_tiles;
    int tb = 0;
    while ((1ll << tb) <= n_tiles) ++tb;
    if (tb < 1) tb = 1;
    hipStream_t s = as_stream(stream);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(keys_scratch);
    hipLaunchKernelGGL((bucket_pass_kernel<true>), dim3(ceil_div(total, 256)), dim3(256), 0, s, C, N, means2d, radii,
                       depths, tile_size, tile_width, tile_height, legacy, nullptr, tile_cursors, shard_offsets, keys,
                       nullptr);
    int np = 1;
    while (np < max_bucket) np <<= 1;
    const size_t lds = (size_t)np * sizeof(unsigned long long);
    hipLaunchKernelGGL(tile_sort_kernel, dim3(T), dim3(256), lds, s, n_tiles, tb, tile_offsets, keys, isect_ids,
                       flatten_ids);
    return check_launch();
}
