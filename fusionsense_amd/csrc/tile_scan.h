// The column scan of the direct binning's [workgroup][tile] table as a device function: tilesort.hip launches it on
// its own, sh.hip launches it together with the SH forward (independent work that fills the machine meanwhile).
#pragma once
#include "common.h"

namespace fsgs {

constexpr int kTrTiles = 8, kTrRows = 512, kTrPer = kTrRows * kTrTiles / 256;
inline int tile_scan_rows_blocks(int T) { return (T + kTrTiles - 1) / kTrTiles; }

// 256 threads; `block` owns tiles block * 8 .. + 7
__device__ __forceinline__ void tile_scan_rows_body(int T, int nb, int32_t *__restrict__ table,
                                                    int32_t *__restrict__ totals, int block) {
    __shared__ int buf[kTrRows][kTrTiles + 1];
    __shared__ int carry[kTrTiles];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t0 = block * kTrTiles;
    const int col = tid & (kTrTiles - 1), row0 = tid / kTrTiles;  // 32 rows x 8 tiles per pass
    const bool col_ok = t0 + col < T;
    if (tid < kTrTiles) carry[tid] = 0;
    for (int b0 = 0; b0 < nb; b0 += kTrRows) {
        const int rows = min(kTrRows, nb - b0);
        int v[kTrPer];
#pragma unroll
        for (int k = 0; k < kTrPer; ++k) {
            const int r = row0 + 32 * k;
            v[k] = (r < rows && col_ok) ? table[(int64_t)(b0 + r) * T + t0 + col] : 0;
        }
#pragma unroll
        for (int k = 0; k < kTrPer; ++k) buf[row0 + 32 * k][col] = v[k];
        __syncthreads();
        // wave w scans tiles 2w, 2w+1: lane l owns rows 8l .. 8l+7 of the chunk
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = 2 * w + cc;
            int x[8], mine = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { x[k] = buf[8 * lane + k][c]; mine += x[k]; }
            int inc = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            int run = carry[c] + inc - mine;
#pragma unroll
            for (int k = 0; k < 8; ++k) { buf[8 * lane + k][c] = run; run += x[k]; }
            const int tot = __shfl(inc, 63, 64);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) carry[c] += tot;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kTrPer; ++k) {
            const int r = row0 + 32 * k;
            if (r < rows && col_ok) table[(int64_t)(b0 + r) * T + t0 + col] = buf[r][col];
        }
        __syncthreads();
    }
    // (write-through: the workgroup that arrives last in scan_rows_sh_pack_kernel reads every tile's total in the launch)
    if (tid < kTrTiles && t0 + tid < T)
        __hip_atomic_store(&totals[t0 + tid], carry[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// isect_offsets from the per-tile totals by ONE workgroup of 256 threads (what tile_offsets_kernel does as a launch of its
// own): offsets[t] = min(sum of totals before t, capacity), offsets[T] = min(total, capacity); total_mapped as there.
__device__ __forceinline__ void tile_offsets_body256(int T, const int32_t *__restrict__ totals,
                                                     int32_t *__restrict__ offsets, int32_t *__restrict__ total_mapped,
                                                     int32_t *__restrict__ tile_order = nullptr) {
    __shared__ int wsum[4];
    __shared__ int ocnt2[32], obase2[32];
    if (tile_order && threadIdx.x < 32) ocnt2[threadIdx.x] = 0;
    if (tile_order) __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (T + 255) / 256;          // consecutive tiles per thread
    const int i0 = tid * per, i1 = min(i0 + per, T);
    const int c = total_mapped ? total_mapped[2] : 0;
    const int cap = c > 0 ? c : 0x7FFFFFFF;
    int mine = 0;
    for (int i = i0; i < i1; ++i) {
        const int n = __hip_atomic_load(&totals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mine += n;
        if (tile_order) atomicAdd(&ocnt2[min(n >> 6, 31)], 1);
    }
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (tile_order) {  // the tiles longest list first (classes of 64 entries), as bucket_offsets_body256 leaves them
        if (tid == 0) {
            int run_o = 0;
            for (int cc = 31; cc >= 0; --cc) { obase2[cc] = run_o; run_o += ocnt2[cc]; }
        }
        __syncthreads();
    }
    int run = inc - mine;
    for (int k = 0; k < w; ++k) run += wsum[k];
    for (int i = i0; i < i1; ++i) {
        const int n = __hip_atomic_load(&totals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        offsets[i] = min(run, cap);
        run += n;
        if (tile_order) tile_order[atomicAdd(&obase2[min(n >> 6, 31)], 1)] = i;
    }
    if (tid == 255) {
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        offsets[T] = min(total, cap);
        if (total_mapped) {
            total_mapped[0] = total;
            __threadfence_system();
            __hip_atomic_store(&total_mapped[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // "the total has landed"
        }
    }
}

struct BucketBook {  // (arguments of bucket_offsets_body256; cursor == nullptr: not the count-free route)
    int32_t *cursor;
    const int32_t *base;
    int32_t *next_base;
    float growth;
    int slack;
    // nullable [T]: the tiles longest list first (classes of 64 entries) — the dispatch order of the in-tile sorts, whose
    // workgroups take as long as their tile is long (the launch is 5-7 rounds of resident workgroups)
    int32_t *tile_order;
};

// The count-free binning route's bookkeeping, ONE workgroup of 256 threads (rides in the SH launch as block 0):
// cursor[t] = the live pairs the fill kernel dropped into tile t's bucket, bucket_base[t] .. [t + 1] = the bucket's room.
//   isect_offsets[t] = exclusive sum of the counts (clamped to the list capacity, as tile_offsets_kernel does),
//   next_base[t]     = exclusive sum of  count * growth + slack  — the room the NEXT frame of this view gets,
//   cursor[t]        = 0 again (this is its last reader in the frame),
//   total_mapped     = [0] the total, [3] 1 if some tile outgrew its bucket (the frame is redone through the exact
//                      two-pass route), then [1] = 1.
__device__ __forceinline__ void bucket_offsets_body256(int T, int32_t *__restrict__ cursor,
                                                       const int32_t *__restrict__ bucket_base,
                                                       int32_t *__restrict__ offsets, int32_t *__restrict__ next_base,
                                                       float growth, int slack, int32_t *__restrict__ total_mapped,
                                                       int32_t *__restrict__ tile_order = nullptr) {
    __shared__ int wsum[4], wsum2[4], over_s[4], tsum[4];
    __shared__ int ocnt[32], obase[32];
    if (tile_order && threadIdx.x < 32) ocnt[threadIdx.x] = 0;
    if (tile_order) __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (T + 255) / 256;
    const int i0 = tid * per, i1 = min(i0 + per, T);
    const int c = total_mapped ? total_mapped[2] : 0;
    const int cap = c > 0 ? c : 0x7FFFFFFF;
    // (a tile that outgrew its bucket holds a valid prefix of unknown length: its list is left EMPTY — the frame is
    // flagged and redone, but nothing downstream may ever read an unwritten word as a Gaussian id)
    int mine = 0, mine2 = 0, over = 0, true_total = 0;
    for (int i = i0; i < i1; ++i) {
        const int n = cursor[i];
        const bool fits = n <= bucket_base[i + 1] - bucket_base[i];
        over |= fits ? 0 : 1;
        mine += fits ? n : 0;
        true_total += n;
        mine2 += (int)((float)n * growth) + slack;
        if (tile_order) atomicAdd(&ocnt[min((fits ? n : 0) >> 6, 31)], 1);
    }
    int inc = mine, inc2 = mine2;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64), o2 = __shfl_up(inc2, d, 64);
        if (lane >= d) { inc += o; inc2 += o2; }
    }
    over = __any(over) ? 1 : 0;
    if (lane == 63) { wsum[w] = inc; wsum2[w] = inc2; over_s[w] = over; }
    __syncthreads();
    if (tile_order) {
        // counting sort of the tiles by list length, longest class first (the class counts are complete: a barrier lies
        // behind them); obase[c] then runs as the class's cursor
        if (tid == 0) {
            int run_o = 0;
            for (int cc = 31; cc >= 0; --cc) { obase[cc] = run_o; run_o += ocnt[cc]; }
        }
        __syncthreads();
    }
    int run = inc - mine, run2 = inc2 - mine2;
    for (int k = 0; k < w; ++k) { run += wsum[k]; run2 += wsum2[k]; }
    for (int i = i0; i < i1; ++i) {
        const int n = cursor[i];
        const bool fits = n <= bucket_base[i + 1] - bucket_base[i];
        offsets[i] = min(run, cap);
        next_base[i] = run2;
        run += fits ? n : 0;
        run2 += (int)((float)n * growth) + slack;
        cursor[i] = 0;
        if (tile_order) tile_order[atomicAdd(&obase[min((fits ? n : 0) >> 6, 31)], 1)] = i;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) true_total += __shfl_xor(true_total, d, 64);
    if (lane == 0) tsum[w] = true_total;
    __syncthreads();
    if (tid == 255) {
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        offsets[T] = min(total, cap);
        next_base[T] = wsum2[0] + wsum2[1] + wsum2[2] + wsum2[3];
        if (total_mapped) {
            total_mapped[0] = tsum[0] + tsum[1] + tsum[2] + tsum[3];  // (the true number of live pairs)
            total_mapped[3] = over_s[0] | over_s[1] | over_s[2] | over_s[3];
            __threadfence_system();
            __hip_atomic_store(&total_mapped[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace fsgs
