// E5 + E6 for the live lists: the (tile | depth) sort as ONE partition by tile plus a sort inside every
// tile's bucket, instead of a 44-bit LSD radix sort over the whole array (the
// reference calls cub::DeviceRadixSort::SortPairs + isect_offset_encode through gsplat 1.0.0
// `isect_tiles(sort=True)` / `isect_offset_encode`, reached from
// /root/reference/dn_splatter/dn_model.py:570-591).
//
// The global sort needed 6 passes x 3 dependent launches (151 us for M = 1.16 M on BASELINE config #2,
// launch- and latency-bound: each pass moves 14 MB).  The high key bits are the tile index, so:
//   1. tile_hist     per block of 8192 keys: histogram over the tiles in LDS -> table[tile][block]
//   2. tile_scan     one workgroup per tile: exclusive prefix over the blocks + the tile's total
//   3. tile_offsets  one workgroup: exclusive scan of the totals = isect_offsets (no offset_encode pass)
//   4. tile_scatter  per block: every key claims the next slot of its tile's bucket (LDS cursor)
//                    and drops  bits(depth) << 32 | id << 4 | quadrant mask  there
//   5. tile_sort     one workgroup per tile: bitonic sort of the bucket's 64-bit words in LDS
// The claiming order inside a block is arbitrary, the in-tile sort on (depth, id) makes the result
// deterministic and equal to a stable sort of the emission order (ties -> ascending flatten id), i.e.
// bit-exact with the radix path and the oracle.
#include <cstdlib>

#include "common.h"
#include "tile_scan.h"

// A/B build switches (make EXTRA=-D...=0); the library reads no environment variable.
#ifndef FSGS_LDS_SPLIT_SORT
#define FSGS_LDS_SPLIT_SORT 1
#endif
#ifndef FSGS_SORT_ONE_TIER
#define FSGS_SORT_ONE_TIER 1
#endif
// the one-tier launch's workgroup size and LDS words per tile (larger tiles are sorted in place: slow)
#ifndef FSGS_SORT_TIER_THREADS
#define FSGS_SORT_TIER_THREADS 1024
#endif
#ifndef FSGS_SORT_TIER_CAP
#define FSGS_SORT_TIER_CAP 8192
#endif

namespace fsgs {

constexpr int kTsThreads = 256;                                  // scan / small-sort workgroup
constexpr int kTpThreads = 1024, kTpItems = 8, kTsBlockKeys = kTpThreads * kTpItems;  // partition kernels: 8192 keys / workgroup

__device__ __forceinline__ int tile_bin(uint64_t key, int tile_bits, int n_tiles) {
    const uint32_t hi = (uint32_t)(key >> 32);
    return (int)(hi >> tile_bits) * n_tiles + (int)(hi & ((1u << tile_bits) - 1u));
}

__global__ void __launch_bounds__(kTpThreads)
tile_hist_kernel(int64_t n, const uint64_t *__restrict__ keys, int tile_bits, int n_tiles, int T, int nb,
                 int32_t *__restrict__ table) {
    extern __shared__ int hist[];
    for (int b = threadIdx.x; b < T; b += kTpThreads) hist[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTsBlockKeys;
#pragma unroll
    for (int k = 0; k < kTpItems; ++k) {
        const int64_t i = base + k * kTpThreads + threadIdx.x;
        if (i < n) atomicAdd(&hist[tile_bin(keys[i], tile_bits, n_tiles)], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < T; b += kTpThreads) table[(int64_t)b * nb + blockIdx.x] = hist[b];
}

// one workgroup per tile: table[tile][0..nb) -> exclusive prefix in place, totals[tile]
__global__ void __launch_bounds__(kTsThreads)
tile_scan_kernel2(int nb, int32_t *__restrict__ table, int32_t *__restrict__ totals) {
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    int32_t *row = table + (int64_t)blockIdx.x * nb;
    for (int base = 0; base < nb; base += kTsThreads) {
        const int i = base + tid;
        const int v = (i < nb) ? row[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        const int carry = carry_s;
        if (i < nb) row[i] = carry + wbase + inc - v;
        __syncthreads();
        if (tid == kTsThreads - 1) carry_s = carry + wbase + inc;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry_s;
}

// The same scan for a table stored [block][tile] (rows written coalesced by the direct-binning count pass of
// isect.hip and read back coalesced by its scatter pass): a workgroup owns 8 tiles, walks the blocks in chunks
// of 512 rows through LDS (32-byte row segments; all of a thread's 16 loads are issued before the first LDS
// write), the exclusive prefixes go back in place, the running totals are carried from chunk to chunk.
__global__ void __launch_bounds__(256)
tile_scan_rows_kernel(int T, int nb, int32_t *__restrict__ table, int32_t *__restrict__ totals) {
    tile_scan_rows_body(T, nb, table, totals, blockIdx.x);
}

// offsets[t] = sum of totals[0..t), offsets[T] = M   (one workgroup)
// total_mapped (nullable): int32[4] in host-mapped pinned memory — [0] <- the TRUE total, [1] <- done flag,
// [2] = capacity given by the host (0 = none): every offset is clamped to it, so that a caller who sized its list
// buffers from an estimate instead of waiting for the total stays inside them (the lists are then truncated and the
// caller, seeing [0] > [2], redoes the frame), [3] reserved.
__global__ void __launch_bounds__(1024)
tile_offsets_kernel(int T, const int32_t *__restrict__ totals, int32_t *__restrict__ offsets,
                    int32_t *__restrict__ total_mapped) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    __shared__ int cap_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) {
        carry_s = 0;
        const int c = total_mapped ? total_mapped[2] : 0;
        cap_s = c > 0 ? c : 0x7FFFFFFF;
    }
    __syncthreads();
    const int cap = cap_s;
    for (int base = 0; base < T; base += 1024) {
        const int i = base + tid;
        const int v = (i < T) ? totals[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        const int carry = carry_s;
        if (i < T) offsets[i] = min(carry + wbase + inc - v, cap);
        __syncthreads();
        if (tid == 1023) carry_s = carry + wbase + inc;
        __syncthreads();
    }
    if (tid == 0) {
        offsets[T] = min(carry_s, cap);
        // (optional) the total straight into host-mapped pinned memory: no device-to-host copy launch for the one
        // number the host waits for
        if (total_mapped) {
            total_mapped[0] = carry_s;
            __threadfence_system();
            __hip_atomic_store(&total_mapped[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // "the total has landed"
        }
    }
}

__global__ void __launch_bounds__(kTpThreads)
tile_scatter_kernel(int64_t n, const uint64_t *__restrict__ keys, const int32_t *__restrict__ payload,
                    int tile_bits, int n_tiles, int T, int nb, const int32_t *__restrict__ table,
                    const int32_t *__restrict__ offsets, uint64_t *__restrict__ buckets) {
    extern __shared__ int cursor[];
    for (int b = threadIdx.x; b < T; b += kTpThreads) cursor[b] = offsets[b] + table[(int64_t)b * nb + blockIdx.x];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTsBlockKeys;
#pragma unroll
    for (int k = 0; k < kTpItems; ++k) {
        const int64_t i = base + k * kTpThreads + threadIdx.x;
        if (i < n) {
            const uint64_t key = keys[i];
            const uint32_t pay = (uint32_t)payload[i];
            const int pos = atomicAdd(&cursor[tile_bin(key, tile_bits, n_tiles)], 1);
            // in-tile sort word: depth bits, then the flatten id (ties -> ascending id), mask in the low bits
            buckets[pos] = ((key & 0xFFFFFFFFull) << 32) | ((uint64_t)(pay & 0x0FFFFFFFu) << 4) | (pay >> 28);
        }
    }
}

// ---- a wave sorts 128 64-bit words in REGISTERS: lane l holds words l and l + 64 of the block -------------
// Every stage of a bitonic merge of span <= 128 pairs a word with the same slot of lane (l ^ mask) — except
// the span-128 mirror (other slot of lane 63 - l) and distance 64 (the lane's own two words) — so the
// exchange is a DPP move (masks 1, 2, 3, 7, 15), a ds_swizzle (4, 8, 16, 31) or a ds_bpermute (32, 63):
// no LDS storage traffic at all, where the LDS version moved 32 bytes per lane per stage through banks
// that the 64-bit stride-2 patterns of the low stages only half use (measured: LDS-bandwidth-bound).
template <int M>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v) {
    if (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    if (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    if (M == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
    if (M == 7) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    if (M == 15) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true); // row_mirror
    if (M < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (M << 10) | 0x1F);    // bit mode: lane ^ M
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63) ^ M) << 2), (int)v);
}
template <int M>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v) {
    return ((uint64_t)lane_xor32<M>((uint32_t)(v >> 32)) << 32) | lane_xor32<M>((uint32_t)v);
}
__device__ __forceinline__ void keep(uint64_t &a, uint64_t p, bool keep_min) {
    a = ((a > p) == keep_min) ? p : a;
}
// both slots against lane (l ^ M); the lane whose bit LOWBIT is clear holds the lower position
template <int M, int LOWBIT>
__device__ __forceinline__ void xstage(uint64_t &e0, uint64_t &e1, int lane) {
    const bool lower = (lane & LOWBIT) == 0;
    const uint64_t p0 = lane_xor64<M>(e0), p1 = lane_xor64<M>(e1);
    keep(e0, p0, lower);
    keep(e1, p1, lower);
}
// the xor stages j = 32 .. 1 that end every merge of span >= 128
__device__ __forceinline__ void merge_tail_32(uint64_t &e0, uint64_t &e1, int lane) {
    xstage<32, 32>(e0, e1, lane); xstage<16, 16>(e0, e1, lane); xstage<8, 8>(e0, e1, lane);
    xstage<4, 4>(e0, e1, lane); xstage<2, 2>(e0, e1, lane); xstage<1, 1>(e0, e1, lane);
}
__device__ __forceinline__ void sort128(uint64_t &e0, uint64_t &e1, int lane) {
    xstage<1, 1>(e0, e1, lane);                                                                    // k = 2
    xstage<3, 2>(e0, e1, lane); xstage<1, 1>(e0, e1, lane);                                         // k = 4
    xstage<7, 4>(e0, e1, lane); xstage<2, 2>(e0, e1, lane); xstage<1, 1>(e0, e1, lane);              // k = 8
    xstage<15, 8>(e0, e1, lane); xstage<4, 4>(e0, e1, lane); xstage<2, 2>(e0, e1, lane);
    xstage<1, 1>(e0, e1, lane);                                                                    // k = 16
    xstage<31, 16>(e0, e1, lane); xstage<8, 8>(e0, e1, lane); xstage<4, 4>(e0, e1, lane);
    xstage<2, 2>(e0, e1, lane); xstage<1, 1>(e0, e1, lane);                                         // k = 32
    xstage<63, 32>(e0, e1, lane); xstage<16, 16>(e0, e1, lane); xstage<8, 8>(e0, e1, lane);
    xstage<4, 4>(e0, e1, lane); xstage<2, 2>(e0, e1, lane); xstage<1, 1>(e0, e1, lane);              // k = 64
    {   // k = 128 mirror: word l <-> word 127 - l = slot 1 of lane 63 - l
        const uint64_t q0 = lane_xor64<63>(e1), q1 = lane_xor64<63>(e0);
        keep(e0, q0, true);
        keep(e1, q1, false);
    }
    merge_tail_32(e0, e1, lane);
}

// One workgroup per tile sorts its bucket with a bitonic network whose compare-exchanges are ALL ascending
// (the first stage of every merge mirrors the upper half).  Buckets with LO < n <= CAP words: every wave
// sorts 128-word blocks in registers (above); merges of span >= 256 do their mirror stage and the stages
// of distance >= 128 in LDS (workgroup barriers), then each wave takes its blocks back into registers
// for the distances 64 .. 1 — for a 512-word bucket that is 6 LDS passes instead of 45, and the last
// pass leaves the registers straight for global memory.  With SPILL, buckets larger than CAP are sorted
// where they lie (L1-bypassing loads/stores, missing words act as +inf without being stored: slow, for
// tiles too dense for LDS — the caller keeps the global radix sort for scenes that dense).
template <int THREADS, int CAP, int LO, bool SPILL>
__device__ __forceinline__ void
tile_sort_body(uint64_t *sk, const int tile_lin, int n_tiles, int tile_bits, const int32_t *__restrict__ offsets,
               uint64_t *__restrict__ buckets, int32_t *__restrict__ payload_out, int64_t *__restrict__ isect_ids_out,
               const int32_t *__restrict__ ends = nullptr, const int32_t *__restrict__ src_offsets = nullptr) {
    // (ends: explicit end of every bucket, for bucket lists with gaps between consecutive entries)
    const int s = offsets[tile_lin], e = ends ? ends[tile_lin] : offsets[tile_lin + 1];
    const int n = e - s;
    if (n <= LO || (CAP > 0 && !SPILL && n > CAP)) return;
    const bool in_lds = CAP > 0 && n <= CAP;
    int np = 1;
    while (np < n) np <<= 1;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = THREADS / 64;
    // (src_offsets: the bucket's words lie at another place than the sorted list goes to — buckets with room to spare,
    // filled without a count pass: fsgs_project_bin_live_fill_sh_pack)
    uint64_t *g = buckets + (src_offsets ? src_offsets[tile_lin] : s);
    auto ld = [&](int i) -> uint64_t {
        return in_lds ? sk[i] : __hip_atomic_load(&g[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto st = [&](int i, uint64_t v) {
        if (in_lds) sk[i] = v;
        else __hip_atomic_store(&g[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto cmpex = [&](int a, int b) {
        if (b < n) {
            const uint64_t ka = ld(a), kb = ld(b);
            if (ka > kb) { st(a, kb); st(b, ka); }
        }
    };
    auto mirror_pair = [&](int i, int k, int &a, int &b) {  // pair i of the mirror stage of merge size k
        const int hk = k >> 1, blk = i / hk, off = i - blk * hk;
        a = blk * k + off; b = blk * k + k - 1 - off;
    };
    auto xor_pair = [&](int i, int j, int &a, int &b) {     // pair i of the stage with partner distance j
        a = ((i & ~(j - 1)) << 1) | (i & (j - 1)); b = a | j;
    };
    const int half = np >> 1;  // pairs per stage
    const int cam = tile_lin / n_tiles, tile = tile_lin - cam * n_tiles;
    const int64_t hi = ((int64_t)cam << (32 + tile_bits)) | ((int64_t)tile << 32);
    auto emit = [&](int pos, uint64_t k) {
        if (pos < n) {
            const uint32_t lo = (uint32_t)(k & 0xFFFFFFFFull);
            payload_out[(int64_t)s + pos] = (int32_t)(((lo & 0xFu) << 28) | (lo >> 4));
            if (isect_ids_out) isect_ids_out[(int64_t)s + pos] = hi | (int64_t)(k >> 32);
        }
    };
    if (in_lds) {
        // 128-word blocks are sorted in registers (missing words are +inf there and in LDS: np <= CAP words)
        const int nblk = np > 128 ? np >> 7 : 1;
        for (int blk = wid; blk < nblk; blk += NW) {
            const int base = blk << 7;
            uint64_t e0 = (base + lane < n) ? g[base + lane] : ~0ull;
            uint64_t e1 = (base + lane + 64 < n) ? g[base + lane + 64] : ~0ull;
            sort128(e0, e1, lane);
            if (np > 128) { sk[base + lane] = e0; sk[base + lane + 64] = e1; }
            else { emit(lane, e0); emit(lane + 64, e1); }
        }
        if (np <= 128) return;
        __syncthreads();
        for (int k = 256; k <= np; k <<= 1) {
            for (int i = tid; i < half; i += THREADS) {  // mirror stage of the merge, across waves: LDS
                int a, b;
                mirror_pair(i, k, a, b);
                const uint64_t ka = sk[a], kb = sk[b];
                if (ka > kb) { sk[a] = kb; sk[b] = ka; }
            }
            __syncthreads();
            for (int j = k >> 2; j >= 128; j >>= 1) {
                for (int i = tid; i < half; i += THREADS) {
                    int a, b;
                    xor_pair(i, j, a, b);
                    const uint64_t ka = sk[a], kb = sk[b];
                    if (ka > kb) { sk[a] = kb; sk[b] = ka; }
                }
                __syncthreads();
            }
            for (int blk = wid; blk < nblk; blk += NW) {  // distances 64 .. 1: back in registers
                const int base = blk << 7;
                uint64_t e0 = sk[base + lane], e1 = sk[base + lane + 64];
                if (e0 > e1) { const uint64_t t = e0; e0 = e1; e1 = t; }
                merge_tail_32(e0, e1, lane);
                if (k < np) { sk[base + lane] = e0; sk[base + lane + 64] = e1; }
                else { emit(base + lane, e0); emit(base + lane + 64, e1); }
            }
            if (k < np) __syncthreads();
        }
        return;
    } else {
        for (int k = 2; k <= np; k <<= 1) {
            for (int i = tid; i < half; i += THREADS) {
                int a, b;
                mirror_pair(i, k, a, b); cmpex(a, b);
            }
            __syncthreads();
            for (int j = k >> 2; j > 0; j >>= 1) {
                for (int i = tid; i < half; i += THREADS) {
                    int a, b;
                    xor_pair(i, j, a, b); cmpex(a, b);
                }
                __syncthreads();
            }
        }
    }
    for (int i = tid; i < n; i += THREADS) emit(i, ld(i));
}

template <int THREADS, int CAP, int LO, bool SPILL>
__global__ void __launch_bounds__(THREADS)
tile_sort_kernel2(int n_tiles, int tile_bits, const int32_t *__restrict__ offsets, uint64_t *__restrict__ buckets,
                  int32_t *__restrict__ payload_out, int64_t *__restrict__ isect_ids_out,
                  int32_t *__restrict__ long_flag, int rel_gate, const int32_t *__restrict__ src_offsets,
                  const int32_t *__restrict__ tile_order) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
    // (tile_order, nullable: the tiles longest first — a workgroup takes as long as its tile is long, and the launch is
    // 5-7 rounds of resident workgroups: its last round should be the short ones)
    const int t = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;
    // (frame statistic for the forward's hand-off, fsgs_bin_live_emit: does ANY tile hold a list of more than rel_gate
    // times the mean length?  The caller zeroes the word; every writer writes the same 1.)
    if (long_flag && threadIdx.x == 0) {
        const int n = offsets[t + 1] - offsets[t];
        if (n > 0 && (int64_t)n * (int)gridDim.x > (int64_t)rel_gate * offsets[gridDim.x]) *long_flag = 1;
    }
    tile_sort_body<THREADS, CAP, LO, SPILL>(sk, t, n_tiles, tile_bits, offsets, buckets, payload_out,
                                            isect_ids_out, nullptr, src_offsets);
}

// (Round 4, measured and removed: one WAVE per bucket of up to 1024 words — 1 / 2 / 4 / 8 blocks of 128 per lane pair,
// sort128 per block, the merges' mirror / block-distance / distance-64 stages as register compares and lane reversals,
// no LDS storage and no barrier, four buckets per 256-thread workgroup.  Bit-identical (the whole binning suite passed)
// and SLOWER: bucket fill + sorts 0.0818 vs 0.0695 ms at config #2, 0.0928 vs 0.0724 ms at config #3 — a wave walks its
// blocks' 57-stage networks one after the other, so a 512-word bucket's critical path is four times the workgroup
// kernel's, where four waves sort one block each; the barriers it removes were not the cost.)
// The same over a list of buckets whose LENGTH lives on the device (`n_sub`: the sub-buckets the depth-slab split
// produced for the few tiles too large for the in-LDS split + sort): a fixed grid strides over them, so a frame
// without such tiles costs a handful of workgroups instead of one early-out workgroup per possible sub-bucket.
template <int THREADS, int CAP, int LO, bool SPILL>
__global__ void __launch_bounds__(THREADS)
tile_sort_strided_kernel(const int32_t *__restrict__ n_sub, int n_tiles, int tile_bits,
                         const int32_t *__restrict__ offsets, const int32_t *__restrict__ ends,
                         uint64_t *__restrict__ buckets, int32_t *__restrict__ payload_out) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
    const int limit = *n_sub;
    for (int b = blockIdx.x; b < limit; b += gridDim.x) {
        tile_sort_body<THREADS, CAP, LO, SPILL>(sk, b, n_tiles, tile_bits, offsets, buckets, payload_out, nullptr, ends);
        __syncthreads();  // (the next bucket reuses the LDS words)
    }
}

constexpr int kTsSmall = 1024, kTsLarge = 8192;  // LDS words per tile: 8 KB x 8 workgroups, 64 KB x 2 per CU
inline int64_t ts_blocks(int64_t n) { return (n + kTsBlockKeys - 1) / kTsBlockKeys; }

// (shared with the direct-binning entry points of isect.hip)
int launch_tile_scan_offsets(int T, int nb, int32_t *table, int32_t *totals, int32_t *isect_offsets, hipStream_t s) {
    hipLaunchKernelGGL(tile_scan_kernel2, dim3(T), dim3(kTsThreads), 0, s, nb, table, totals);
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, s, T, totals, isect_offsets, nullptr);
    return check_launch();
}

int launch_tile_offsets(int T, const int32_t *totals, int32_t *isect_offsets, int32_t *total_mapped, hipStream_t s) {
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, s, T, totals, isect_offsets, total_mapped);
    return check_launch();
}

int launch_tile_scan_rows_offsets(int T, int nb, int32_t *table, int32_t *totals, int32_t *isect_offsets,
                                  int32_t *total_mapped, hipStream_t s) {
    hipLaunchKernelGGL(tile_scan_rows_kernel, dim3((T + kTrTiles - 1) / kTrTiles), dim3(256), 0, s, T, nb, table, totals);
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, s, T, totals, isect_offsets, total_mapped);
    return check_launch();
}

// ---- depth slabs: buckets too large for the LDS sorts are split by depth first --------------------------------
// A tile's bucket of n > kSplitMin words is partitioned into S = ceil(n / kSplitTarget) slabs by
// slab(depth) = floor((bits(depth) - min) * S / (max - min + 1))  (monotone in depth, equal depths share a slab),
// every slab is then sorted on (depth, id) like a small tile and the slabs lie in order: the tile is sorted.
// One more pass over the words (24 B / word) instead of the 6 radix passes (32 B / pair each) that scenes with
// thousands of pairs per tile needed.  Slabs that still exceed the LDS tiers (skewed depths) are sorted in place.
constexpr int kSplitMin = 1024, kSplitTarget = 512, kSplitMaxSlabs = 2048, kSplitThreads = 1024;

__device__ __host__ inline int split_slabs(int n, int target = kSplitTarget) {
    if (n <= kSplitMin) return 1;
    const int s = (n + target - 1) / target;
    return s < kSplitMaxSlabs ? s : kSplitMaxSlabs;
}

// sub_base[t] = number of sub-buckets in front of tile t, sub_base[T] = their total   (one workgroup)
__global__ void __launch_bounds__(1024)
split_base_kernel(int T, const int32_t *__restrict__ offsets, int32_t *__restrict__ sub_base, int lds_cap,
                  int target) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        const int i = base + tid;
        int v = 0;
        if (i < T) {
            const int n_i = offsets[i + 1] - offsets[i];
            v = (n_i <= lds_cap) ? 0 : split_slabs(n_i, target);  // (tiles the in-LDS split + sort has taken: none)
        }
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        const int carry = carry_s;
        if (i < T) sub_base[i] = carry + wbase + inc - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + wbase + inc;
        __syncthreads();
    }
    if (tid == 0) sub_base[T] = carry_s;
}

// one workgroup per tile: words of the tile's bucket -> `out`, grouped by slab; sub_offsets[sub_base[t] + k] =
// first word of slab k; the last tile's workgroup also closes the list (entries up to max_sub = total words).
// The slabs are cut at QUANTILES of a 2048-bin depth histogram of the bucket (bin b belongs to slab
// floor(words in front of b / kSplitTarget)): depths cluster on surfaces, equal-width slabs left most words in a
// few slabs that only the large sort tier could take.
constexpr int kSplitBins = 2048;
__global__ void __launch_bounds__(kSplitThreads)
split_slabs_kernel(int T, const int32_t *__restrict__ offsets, const int32_t *__restrict__ sub_base,
                   const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int32_t *__restrict__ sub_offsets,
                   int32_t *__restrict__ sub_ends, int max_sub) {
    __shared__ unsigned int red_min[kSplitThreads / 64], red_max[kSplitThreads / 64];
    __shared__ int excl[kSplitBins];    // words per depth bin, then words in front of the bin
    __shared__ int cursor[kSplitMaxSlabs];
    __shared__ int wsum[kSplitThreads / 64];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int s = offsets[t], e = offsets[t + 1], n = e - s;
    const int sb = sub_base[t], S = sub_base[t + 1] - sb;
    if (t == T - 1) {
        const int total = offsets[T];
        for (int i = sub_base[T] + tid; i <= max_sub; i += kSplitThreads) sub_offsets[i] = total;
    }
    if (S == 0) return;  // (sorted by tile_split_sort_lds_kernel)
    if (S == 1) {
        for (int i = tid; i < n; i += kSplitThreads) out[s + i] = in[s + i];
        if (tid == 0) { sub_offsets[sb] = s; sub_ends[sb] = e; }
        return;
    }
    const int tgt = (n + S - 1) / S;  // words per slab (the caller's target, or more where S hit kSplitMaxSlabs)
    // One workgroup walks the whole bucket three times; the largest tile of a dense frame (10^5 words) sets the
    // kernel's duration, so every pass keeps eight independent loads per thread in flight (1024 threads).
    constexpr int U = 8;
    // depth range of the bucket
    unsigned int lo = 0xFFFFFFFFu, hi = 0u;
    for (int i0 = tid; i0 < n; i0 += U * kSplitThreads) {
        uint64_t wd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) wd[u] = (i0 + u * kSplitThreads < n) ? in[s + i0 + u * kSplitThreads] : in[s + i0];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned int d = (unsigned int)(wd[u] >> 32);
            lo = min(lo, d); hi = max(hi, d);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, (unsigned int)__shfl_xor((int)lo, d, 64));
        hi = max(hi, (unsigned int)__shfl_xor((int)hi, d, 64));
    }
    if (lane == 0) { red_min[w] = lo; red_max[w] = hi; }
    for (int k = tid; k < kSplitBins; k += kSplitThreads) excl[k] = 0;
    for (int k = tid; k < S; k += kSplitThreads) cursor[k] = e;  // slabs no bin starts stay empty at the end
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSplitThreads / 64; ++k) { lo = min(lo, red_min[k]); hi = max(hi, red_max[k]); }
    const unsigned long long span = (unsigned long long)(hi - lo) + 1ull;
    auto bin_of = [&](unsigned int d) -> int {
        return (int)(((unsigned long long)(d - lo) * (unsigned long long)kSplitBins) / span);
    };
    for (int i0 = tid; i0 < n; i0 += U * kSplitThreads) {
        uint64_t wd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) wd[u] = (i0 + u * kSplitThreads < n) ? in[s + i0 + u * kSplitThreads] : 0ull;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * kSplitThreads < n) atomicAdd(&excl[bin_of((unsigned int)(wd[u] >> 32))], 1);
    }
    __syncthreads();
    // exclusive scan of the 2048 bin counts: kSplitBins / kSplitThreads per thread
    constexpr int BPT = kSplitBins / kSplitThreads;
    int v[BPT], mine = 0;
#pragma unroll
    for (int k = 0; k < BPT; ++k) { v[k] = excl[BPT * tid + k]; mine += v[k]; }
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int run = inc - mine;
    for (int k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
    for (int k = 0; k < BPT; ++k) { excl[BPT * tid + k] = run; run += v[k]; }
    __syncthreads();
    // bin b opens the slabs in (slab of the bin in front of it, slab(b)]: their first word is the bin's first word
    for (int b = tid; b < kSplitBins; b += kSplitThreads) {
        const int sl = min(excl[b] / tgt, S - 1);
        const int before = (b == 0) ? -1 : min(excl[b - 1] / tgt, S - 1);
        for (int k = before + 1; k <= sl; ++k) cursor[k] = s + excl[b];
    }
    __syncthreads();
    for (int k = tid; k < S; k += kSplitThreads) {
        sub_offsets[sb + k] = cursor[k];
        sub_ends[sb + k] = (k + 1 < S) ? cursor[k + 1] : e;  // (explicit: tiles without sub-buckets may lie in between)
    }
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += U * kSplitThreads) {
        uint64_t wd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) wd[u] = (i0 + u * kSplitThreads < n) ? in[s + i0 + u * kSplitThreads] : 0ull;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * kSplitThreads < n) {
                const int sl = min(excl[bin_of((unsigned int)(wd[u] >> 32))] / tgt, S - 1);
                out[atomicAdd(&cursor[sl], 1)] = wd[u];
            }
        }
    }
}

// ---- depth slabs in LDS: split + sort of a whole tile in ONE workgroup (round 3) ---------------------------------
// Dense scenes (BASELINE configs #4 / #5: thousands of live pairs per tile) went bucket -> split_slabs_kernel (three
// reads of the bucket, one scattered 8-byte write per word to a second global buffer) -> tile_sort (read + sort +
// write): 1.07 ms of a 6.3 ms step at 6 M Gaussians.  A bucket of up to kLdsCap words fits the LDS of one workgroup
// together with its depth histogram, so the same algorithm runs without the intermediate buffer: every thread keeps
// its 8 words in registers, the 2048-bin histogram of the (linearised) depth and its scan give quantile slabs of
// ~kLdsTarget words, the words are scattered into LDS slab by slab, and every WAVE sorts whole slabs of <= 128 words
// in registers (sort128: no barrier) and writes the payload from there.  Slabs that a depth cluster pushed beyond 128
// words are sorted by the whole workgroup in LDS afterwards.  The slab of a word is a monotone function of its depth
// and the words of a slab are sorted on (depth, id): the result is the sorted tile, bit for bit what the bitonic
// tiers produce.
#ifndef FSGS_LDS_TARGET
#define FSGS_LDS_TARGET 80
#endif
constexpr int kLdsCap = 8192, kLdsThreads = 1024, kLdsItems = kLdsCap / kLdsThreads, kLdsTarget = FSGS_LDS_TARGET;
#ifndef FSGS_LDS_COARSE
#define FSGS_LDS_COARSE 5120
#endif
constexpr int kLdsCoarse = FSGS_LDS_COARSE;  // words per coarse slab of a tile beyond kLdsCap (bins of a 10^5-word tile hold ~50)
constexpr int kLdsMaxSlabs = kLdsCap / kLdsTarget + 2;

struct LdsSplitShared {
    int hist[kSplitBins];            // words per depth bin, then words in front of the bin
    int start[kLdsMaxSlabs + 1];     // first word of every slab; start[S] = n
    int cursor[kLdsMaxSlabs];
    unsigned int red_min[kLdsThreads / 64], red_max[kLdsThreads / 64];
    int wsum[kLdsThreads / 64];
};

__device__ __forceinline__ void
lds_split_sort_body(uint64_t *sk, const int s, const int n, const uint64_t *__restrict__ buckets,
                    int32_t *__restrict__ payload_out) {
    LdsSplitShared &L = *reinterpret_cast<LdsSplitShared *>(sk + kLdsCap);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (n <= 0 || n > kLdsCap) return;  // (larger buckets: the global depth-slab split first, launch_split_sort)
    auto emit = [&](int pos, uint64_t k) {
        const uint32_t lo32 = (uint32_t)(k & 0xFFFFFFFFull);
        payload_out[(int64_t)s + pos] = (int32_t)(((lo32 & 0xFu) << 28) | (lo32 >> 4));
    };
    const uint64_t *g = buckets + s;
    if (n <= 128) {  // one wave, one register sort
        if (w == 0) {
            uint64_t e0 = lane < n ? g[lane] : ~0ull, e1 = lane + 64 < n ? g[lane + 64] : ~0ull;
            sort128(e0, e1, lane);
            if (lane < n) emit(lane, e0);
            if (lane + 64 < n) emit(lane + 64, e1);
        }
        return;
    }
    // the thread's words (all loads in flight together) and the bucket's depth range
    uint64_t word[kLdsItems];
    unsigned int lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (int k = 0; k < kLdsItems; ++k) {
        const int i = tid + k * kLdsThreads;
        word[k] = i < n ? g[i] : ~0ull;
    }
#pragma unroll
    for (int k = 0; k < kLdsItems; ++k) {
        if (tid + k * kLdsThreads < n) {
            const unsigned int d = (unsigned int)(word[k] >> 32);
            lo = min(lo, d); hi = max(hi, d);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, (unsigned int)__shfl_xor((int)lo, d, 64));
        hi = max(hi, (unsigned int)__shfl_xor((int)hi, d, 64));
    }
    if (lane == 0) { L.red_min[w] = lo; L.red_max[w] = hi; }
    for (int k = tid; k < kSplitBins; k += kLdsThreads) L.hist[k] = 0;
    const int S = (n - 1) / kLdsTarget + 1;  // <= kLdsMaxSlabs - 1
    for (int k = tid; k <= S; k += kLdsThreads) L.start[k] = n;  // slabs no bin opens stay empty at the end
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kLdsThreads / 64; ++k) { lo = min(lo, L.red_min[k]); hi = max(hi, L.red_max[k]); }
    // bin = floor((d - lo) * 2048 / span) in float: rounding cannot break monotonicity (u32 -> f32 conversion, a
    // product with a positive constant and the truncation are all non-decreasing), and only monotonicity matters
    const float scale = (float)kSplitBins / ((float)(hi - lo) + 1.0f);
    auto bin_of = [&](unsigned int d) -> int {
        return min((int)(__uint2float_rz(d - lo) * scale), kSplitBins - 1);
    };
    int bin[kLdsItems];
#pragma unroll
    for (int k = 0; k < kLdsItems; ++k) {
        bin[k] = 0;
        if (tid + k * kLdsThreads < n) {
            bin[k] = bin_of((unsigned int)(word[k] >> 32));
            atomicAdd(&L.hist[bin[k]], 1);
        }
    }
    __syncthreads();
    {   // exclusive scan of the 2048 bin counts: two per thread
        const int v0 = L.hist[2 * tid], v1 = L.hist[2 * tid + 1];
        int inc = v0 + v1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) L.wsum[w] = inc;
        __syncthreads();
        int run = inc - (v0 + v1);
        for (int k = 0; k < w; ++k) run += L.wsum[k];
        L.hist[2 * tid] = run;
        L.hist[2 * tid + 1] = run + v0;
    }
    __syncthreads();
    // bin b opens the slabs in (slab of the bin in front of it, slab(b)]: their first word is the bin's first word
    for (int b = tid; b < kSplitBins; b += kLdsThreads) {
        const int sl = min(L.hist[b] / kLdsTarget, S - 1);
        const int before = (b == 0) ? -1 : min(L.hist[b - 1] / kLdsTarget, S - 1);
        for (int k = before + 1; k <= sl; ++k) L.start[k] = L.hist[b];
    }
    __syncthreads();
    for (int k = tid; k < S; k += kLdsThreads) L.cursor[k] = L.start[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kLdsItems; ++k) {
        if (tid + k * kLdsThreads < n) {
            const int sl = min(L.hist[bin[k]] / kLdsTarget, S - 1);
            sk[atomicAdd(&L.cursor[sl], 1)] = word[k];
        }
    }
    __syncthreads();
    // every wave: whole slabs of <= 128 words in registers, straight to the payload
    bool any_big = false;
    for (int j = w; j < S; j += kLdsThreads / 64) {
        const int a = L.start[j], m = L.start[j + 1] - a;
        if (m > 128) { any_big = true; continue; }
        if (m <= 0) continue;
        uint64_t e0 = lane < m ? sk[a + lane] : ~0ull, e1 = lane + 64 < m ? sk[a + lane + 64] : ~0ull;
        if (m > 1) sort128(e0, e1, lane);
        if (lane < m) emit(a + lane, e0);
        if (lane + 64 < m) emit(a + lane + 64, e1);
    }
    if (!__syncthreads_or(any_big ? 1 : 0)) return;
    // (rare: a depth cluster left a slab of more than 128 words) the workgroup sorts it where it lies in LDS
    for (int j = 0; j < S; ++j) {
        const int a = L.start[j], m = L.start[j + 1] - a;
        if (m <= 128) continue;  // (uniform over the workgroup)
        int np = 256;
        while (np < m) np <<= 1;
        const int half = np >> 1;
        uint64_t *q = sk + a;
        auto cmpex = [&](int x, int y) {
            if (y < m) {
                const uint64_t kx = q[x], ky = q[y];
                if (kx > ky) { q[x] = ky; q[y] = kx; }
            }
        };
        for (int k = 2; k <= np; k <<= 1) {
            for (int i = tid; i < half; i += kLdsThreads) {
                const int hk = k >> 1, blk = i / hk, off = i - blk * hk;
                cmpex(blk * k + off, blk * k + k - 1 - off);
            }
            __syncthreads();
            for (int d = k >> 2; d > 0; d >>= 1) {
                for (int i = tid; i < half; i += kLdsThreads) {
                    const int x = ((i & ~(d - 1)) << 1) | (i & (d - 1));
                    cmpex(x, x | d);
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < m; i += kLdsThreads) emit(a + i, q[i]);
        __syncthreads();
    }
}

// one workgroup per tile of the frame ...
__global__ void __launch_bounds__(kLdsThreads)
tile_split_sort_lds_kernel(const int32_t *__restrict__ offsets, const uint64_t *__restrict__ buckets,
                           int32_t *__restrict__ payload_out) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];  // kLdsCap words, then LdsSplitShared
    const int t = blockIdx.x;
    lds_split_sort_body(sk, offsets[t], offsets[t + 1] - offsets[t], buckets, payload_out);
}

// ... and a fixed grid striding over the coarse depth slabs (~kLdsCoarse words each) that split_slabs_kernel cut the
// tiles beyond kLdsCap words into; their number lives on the device (sub_base[T])
__global__ void __launch_bounds__(kLdsThreads)
slab_split_sort_lds_kernel(const int32_t *__restrict__ n_sub, const int32_t *__restrict__ starts,
                           const int32_t *__restrict__ ends, const uint64_t *__restrict__ words,
                           int32_t *__restrict__ payload_out) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
    const int limit = *n_sub;
    for (int b = blockIdx.x; b < limit; b += gridDim.x) {
        lds_split_sort_body(sk, starts[b], ends[b] - starts[b], words, payload_out);
        __syncthreads();  // (the next slab reuses the LDS words)
    }
}

int launch_tile_sort_tiers(int T, int n_tiles, int tile_bits, const int32_t *isect_offsets, uint64_t *buckets,
                           int32_t *payload_sorted, int64_t *isect_ids_sorted, hipStream_t s,
                           int32_t *long_flag = nullptr, int rel_gate = 0, const int32_t *src_offsets = nullptr,
                           const int32_t *tile_order = nullptr);

// buckets -> (split by depth where large) buckets2 -> sorted payload.  scratch: [sub_base i32: T + 1][sub_offsets i32: max_sub + 1]
// lds_cap > 0: tiles of up to lds_cap words have been sorted by tile_split_sort_lds_kernel and yield no sub-buckets here.
int launch_split_sort(int T, int n_tiles, int tile_bits, const int32_t *isect_offsets, int64_t n_live,
                      const uint64_t *buckets, uint64_t *buckets2, int32_t *scratch, int32_t *payload_sorted,
                      hipStream_t s) {
    constexpr int use_lds = FSGS_LDS_SPLIT_SORT;  // (build macro: 0 = the round-2 route through global sub-buckets)
    const int target = use_lds ? kLdsCoarse : kSplitTarget;
    const int max_sub = (int)(T + n_live / target + 1);
    int32_t *sub_base = scratch, *sub_offsets = scratch + (T + 1), *sub_ends = sub_offsets + (max_sub + 1);
    const int lds_cap = use_lds ? kLdsCap : 0;
    const size_t lds = (size_t)kLdsCap * 8 + sizeof(LdsSplitShared);
    if (use_lds) {
        if (const int rc = ensure_dynamic_lds<&tile_split_sort_lds_kernel>(lds)) return rc;
        hipLaunchKernelGGL(tile_split_sort_lds_kernel, dim3(T), dim3(kLdsThreads), lds, s, isect_offsets, buckets,
                           payload_sorted);
    }
    hipLaunchKernelGGL(split_base_kernel, dim3(1), dim3(1024), 0, s, T, isect_offsets, sub_base, lds_cap, target);
    hipLaunchKernelGGL(split_slabs_kernel, dim3(T), dim3(kSplitThreads), 0, s, T, isect_offsets, sub_base, buckets,
                       buckets2, sub_offsets, sub_ends, max_sub);
    int rc = check_launch();
    if (rc != FSGS_OK) return rc;
    if (!use_lds) return launch_tile_sort_tiers(max_sub, n_tiles, tile_bits, sub_offsets, buckets2, payload_sorted, nullptr, s);
    // what is left are the coarse slabs of the tiles beyond lds_cap words (their number: sub_base[T], on the device):
    // each through the same in-LDS split + sort; a slab a depth cluster pushed beyond lds_cap is sorted where it lies
    if (const int rc2 = ensure_dynamic_lds<&slab_split_sort_lds_kernel>(lds)) return rc2;
    if (const int rc2 = ensure_dynamic_lds<&tile_sort_strided_kernel<1024, kTsLarge, kLdsCap, true>>(kTsLarge * 8)) return rc2;
    const int grid = max_sub < 512 ? max_sub : 512;
    hipLaunchKernelGGL(slab_split_sort_lds_kernel, dim3(grid), dim3(kLdsThreads), lds, s, sub_base + T, sub_offsets,
                       sub_ends, buckets2, payload_sorted);
    hipLaunchKernelGGL((tile_sort_strided_kernel<1024, kTsLarge, kLdsCap, true>), dim3(grid < 256 ? grid : 256),
                       dim3(1024), kTsLarge * 8, s, sub_base + T, n_tiles, tile_bits, sub_offsets, sub_ends, buckets2,
                       payload_sorted);
    return check_launch();
}

// [sub_base: T + 1] [sub_offsets: max_sub + 1] [sub_ends: max_sub + 1], max_sub = T + n_live / target + 1 (sized for the
// smaller of the two targets)
int64_t split_scratch_ints(int T, int64_t n_live) { return (int64_t)(T + 1) + 2 * ((T + n_live / kSplitTarget + 1) + 1); }

int launch_tile_sort_tiers(int T, int n_tiles, int tile_bits, const int32_t *isect_offsets, uint64_t *buckets,
                           int32_t *payload_sorted, int64_t *isect_ids_sorted, hipStream_t s, int32_t *long_flag,
                           int rel_gate, const int32_t *src_offsets, const int32_t *tile_order) {
    if (const int rc = ensure_dynamic_lds<&tile_sort_kernel2<1024, kTsLarge, kTsSmall, true>>(kTsLarge * 8)) return rc;
    constexpr int one_tier = FSGS_SORT_ONE_TIER;  // (build macro)
    if (one_tier && T <= 8192) {
        // a moderate number of buckets: EVERY bucket through the 1024-thread kernel in one launch — the small
        // buckets' workgroups (12 of their 16 waves have nothing to do and only meet the barriers) run beside the
        // few long ones instead of in a launch of their own in front of them: 15 + 31 -> 40 us at config #2.
        // With tens of thousands of buckets (large images, the depth-slab path) the 64 KB of LDS per workgroup would
        // limit the small ones to 2 per CU: there the two tiers stay.
        constexpr int TT = FSGS_SORT_TIER_THREADS, TC = FSGS_SORT_TIER_CAP;
        if (const int rc = ensure_dynamic_lds<&tile_sort_kernel2<TT, TC, 0, true>>(TC * 8)) return rc;
        hipLaunchKernelGGL((tile_sort_kernel2<TT, TC, 0, true>), dim3(T), dim3(TT), TC * 8, s, n_tiles,
                           tile_bits, isect_offsets, buckets, payload_sorted, isect_ids_sorted, long_flag, rel_gate,
                           src_offsets, tile_order);
        return check_launch();
    }
    // two size tiers, each skipping the tiles of the other (an early-out workgroup costs ~2 ns)
    hipLaunchKernelGGL((tile_sort_kernel2<256, kTsSmall, 0, false>), dim3(T), dim3(256), kTsSmall * 8, s, n_tiles,
                       tile_bits, isect_offsets, buckets, payload_sorted, isect_ids_sorted, long_flag, rel_gate,
                       src_offsets, tile_order);
    hipLaunchKernelGGL((tile_sort_kernel2<1024, kTsLarge, kTsSmall, true>), dim3(T), dim3(1024), kTsLarge * 8, s,
                       n_tiles, tile_bits, isect_offsets, buckets, payload_sorted, isect_ids_sorted, (int32_t *)nullptr, 0,
                       src_offsets, tile_order);
    return check_launch();
}

}  // namespace fsgs

using namespace fsgs;

// at most this many tiles (cameras x tiles) are handled: the per-block tile histogram lives in LDS
extern "C" int fsgs_tile_sort_max_tiles(void) { return 32768; }

// scratch layout: [table i32: T * nb] [totals i32: T] [buckets u64: n]
extern "C" size_t fsgs_tile_sort_scratch_bytes(int64_t n, int n_tiles_total) {
    const int64_t nb = ts_blocks(n > 0 ? n : 1);
    size_t bytes = ((size_t)n_tiles_total * nb + n_tiles_total) * sizeof(int32_t);
    bytes = (bytes + 15) & ~(size_t)15;
    return bytes + (size_t)(n > 0 ? n : 1) * sizeof(uint64_t) + 64;
}

extern "C" int fsgs_tile_sort(int64_t n, const int64_t *isect_ids, const int32_t *payload, int C, int tile_width,
                              int tile_height, int32_t *isect_offsets /*[C*th*tw + 1]*/, int32_t *payload_sorted,
                              int64_t *isect_ids_sorted /*nullable*/, void *scratch, size_t scratch_bytes,
                              fsgs_stream_t stream) {
    if (n < 0 || C < 1 || tile_width < 1 || tile_height < 1 || !isect_offsets) return FSGS_EINVAL;
    const int n_tiles = tile_width * tile_height;
    const int64_t T64 = (int64_t)C * n_tiles;
    if (T64 > fsgs_tile_sort_max_tiles() || n > 0x7FFFFFF0ll) return FSGS_EINVAL;
    const int T = (int)T64;
    hipStream_t s = as_stream(stream);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(isect_offsets, 0, (size_t)(T + 1) * sizeof(int32_t), s);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
        return FSGS_OK;
    }
    if (!isect_ids || !payload || !payload_sorted || !scratch) return FSGS_EINVAL;
    if (scratch_bytes < fsgs_tile_sort_scratch_bytes(n, T)) return FSGS_ESCRATCH;
    int tb = 0;
    while ((1ll << tb) <= n_tiles) ++tb;
    if (tb < 1) tb = 1;
    const int nb = (int)ts_blocks(n);
    char *p = reinterpret_cast<char *>(scratch);
    int32_t *table = reinterpret_cast<int32_t *>(p);
    int32_t *totals = table + (size_t)T * nb;
    const size_t off = (((size_t)T * nb + T) * sizeof(int32_t) + 15) & ~(size_t)15;
    uint64_t *buckets = reinterpret_cast<uint64_t *>(p + off);
    const uint64_t *keys = reinterpret_cast<const uint64_t *>(isect_ids);
    const size_t lds = (size_t)T * sizeof(int);
    hipLaunchKernelGGL(tile_hist_kernel, dim3(nb), dim3(kTpThreads), lds, s, n, keys, tb, n_tiles, T, nb, table);
    launch_tile_scan_offsets(T, nb, table, totals, isect_offsets, s);
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(nb), dim3(kTpThreads), lds, s, n, keys, payload, tb, n_tiles, T, nb,
                       table, isect_offsets, buckets);
    return launch_tile_sort_tiers(T, n_tiles, tb, isect_offsets, buckets, payload_sorted, isect_ids_sorted, s);
}
