// E4 + E6: Gaussian<->tile intersection (count, emit) and per-tile offset encoding
// (SURVEY.md §8a-5, 8a-7).  Replaces gsplat 1.0.0 `isect_tiles` / `isect_offset_encode`
// (new path, /root/reference/dn_splatter/dn_model.py:570-591) and the legacy
// `map_gaussian_to_intersects` / `get_tile_bin_edges` behind gsplat.rasterize_gaussians
// (dn_model.py:644-653).  Integer work: every output is bit-exact against the oracle.
#include "common.h"
#include "adam_body.h"
#include <cstdlib>
#include "cull.h"
#include "tile_scan.h"
#include "project_math.h"
#include "scan.h"

// A/B build switch (make EXTRA=-DFSGS_ISECT_PER_THREAD=1): one thread per Gaussian in the drop-in path's live count / emit
// passes instead of the flattened pair enumeration.  The library reads no environment variable.
// count-free route: threads per Gaussian of a workgroup, by the workgroup's Gaussians (bin_threads).  Config #3 (256):
// x2 0.527, x4 0.525 ms per step against 0.538; config #2 (512): x2 0.501 against 0.496 (586 workgroups of 1024 threads
// are not all resident).
#ifndef FSGS_FILL_MULT_256
#define FSGS_FILL_MULT_256 4
#endif
#ifndef FSGS_FILL_MULT_512
#define FSGS_FILL_MULT_512 1
#endif
#ifndef FSGS_ISECT_PER_THREAD
#define FSGS_ISECT_PER_THREAD 0
#endif

namespace fsgs {

struct TileRect {
    int x0, y0, x1, y1;  // [min, max) in tile units
};

// New path: floor/ceil.  Legacy: C-cast truncation of (c - r) and (c + r + 1).  Negative
// values are clamped BEFORE the integer conversion (CUDA relies on saturating casts).
__device__ __forceinline__ TileRect tile_rect(float mx, float my, int radius, int tile_size, int tw,
                                              int th, int legacy) {
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
    TileRect r;
    r.x0 = (int)fminf(fmaxf(lx, 0.f), (float)tw);
    r.y0 = (int)fminf(fmaxf(ly, 0.f), (float)th);
    r.x1 = (int)fminf(fmaxf(hx, 0.f), (float)tw);
    r.y1 = (int)fminf(fmaxf(hy, 0.f), (float)th);
    return r;
}

// Also counts (into *rule_diff, when given) the Gaussians whose rectangle differs under the
// OTHER bbox rule: zero means the legacy normal pass may reuse this frame's sorted lists.
__global__ void __launch_bounds__(256)
isect_count_kernel(int64_t total, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                   int tile_size, int tw, int th, int legacy, int32_t *__restrict__ tiles_per_gauss,
                   unsigned long long *__restrict__ rule_diff) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int r = radii[idx];
    int cnt = 0;
    if (r > 0) {
        const float2 m = reinterpret_cast<const float2 *>(means2d)[idx];
        const TileRect t = tile_rect(m.x, m.y, r, tile_size, tw, th, legacy);
        cnt = (t.x1 - t.x0) * (t.y1 - t.y0);
        if (rule_diff) {
            const TileRect o = tile_rect(m.x, m.y, r, tile_size, tw, th, !legacy);
            if (o.x0 != t.x0 || o.y0 != t.y0 || o.x1 != t.x1 || o.y1 != t.y1) atomicAdd(rule_diff, 1ull);
        }
    }
    tiles_per_gauss[idx] = cnt;
}

// One thread per (camera, Gaussian) writes its row-major tile list.  Gaussians covering many
// tiles (>= kWideTiles) are deferred to the whole wave, which writes them 64 tiles at a time
// so that one large splat does not serialise its wave.
constexpr int kWideTiles = 128;

__global__ void __launch_bounds__(256)
isect_emit_kernel(int C, int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                  const float *__restrict__ depths, const int64_t *__restrict__ cum_tiles,
                  int tile_size, int tw, int th, int legacy, int tile_bits,
                  int64_t *__restrict__ isect_ids, int32_t *__restrict__ flatten_ids) {
    const int64_t total = (int64_t)C * N;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    TileRect t = {0, 0, 0, 0};
    int64_t key_hi_base = 0, depth_bits = 0, start = 0;
    int cnt = 0;
    if (idx < total) {
        const int r = radii[idx];
        if (r > 0) {
            const float2 m = reinterpret_cast<const float2 *>(means2d)[idx];
            t = tile_rect(m.x, m.y, r, tile_size, tw, th, legacy);
            cnt = (t.x1 - t.x0) * (t.y1 - t.y0);
            const int64_t c = idx / N;
            key_hi_base = c << (32 + tile_bits);
            depth_bits = (int64_t)(uint32_t)__float_as_int(depths[idx]);
            start = (idx == 0) ? 0 : cum_tiles[idx - 1];
        }
    }
    const bool wide = cnt >= kWideTiles;
    if (cnt > 0 && !wide) {
        int64_t o = start;
        for (int y = t.y0; y < t.y1; ++y)
            for (int x = t.x0; x < t.x1; ++x) {
                isect_ids[o] = key_hi_base | ((int64_t)(y * tw + x) << 32) | depth_bits;
                flatten_ids[o] = (int32_t)idx;
                ++o;
            }
    }
    // wave-cooperative path for wide splats
    unsigned long long wide_mask = __ballot(wide);
    while (wide_mask) {
        const int src = __ffsll((long long)wide_mask) - 1;
        wide_mask &= wide_mask - 1;
        const int x0 = __shfl(t.x0, src, 64), y0 = __shfl(t.y0, src, 64);
        const int x1 = __shfl(t.x1, src, 64);
        const int n = __shfl(cnt, src, 64);
        const int64_t kb = __shfl(key_hi_base, src, 64), db = __shfl(depth_bits, src, 64);
        const int64_t st = __shfl(start, src, 64);
        const int32_t gid = (int32_t)__shfl((long long)idx, src, 64);
        const int wdt = x1 - x0;
        for (int k = lane; k < n; k += 64) {
            const int y = y0 + k / wdt, x = x0 + k % wdt;
            isect_ids[st + k] = kb | ((int64_t)(y * tw + x) << 32) | db;
            flatten_ids[st + k] = gid;
        }
    }
}

// ---- live emission (fused get_outputs node) -------------------------------------------------------
// Same tile rectangles as above, but a (Gaussian, tile) pair is only counted / emitted if the
// Gaussian can reach at least one 8x8 quadrant of the tile at alpha >= 1/255 (cull.h).  The 4-bit
// quadrant mask rides in the payload's top bits (flatten id < 2^28), so the separate list-order
// mask kernel and ~3/4 of the sort, scan and compaction traffic disappear.  tiles_per_gauss still
// reports gsplat's full rectangle count.  EMIT = false: count pass; EMIT = true: emit pass.
template <bool EMIT>
__global__ void __launch_bounds__(256)
isect_live_kernel(int C, int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                  const float *__restrict__ depths, const float *__restrict__ conics,
                  const float *__restrict__ opacities, const int64_t *__restrict__ cum_live, int tw, int th,
                  int tile_bits, int32_t *__restrict__ tiles_per_gauss, int32_t *__restrict__ live_per_gauss,
                  int64_t *__restrict__ isect_ids, int32_t *__restrict__ payload) {
    const int64_t total = (int64_t)C * N;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    TileRect t = {0, 0, 0, 0};
    int cnt = 0;
    float mx = 0.f, my = 0.f, op = 0.f, ca = 0.f, cb = 0.f, cc = 0.f;
    int64_t key_hi_base = 0, depth_bits = 0, start = 0;
    if (idx < total) {
        const int r = radii[idx];
        if (r > 0) {
            const float2 m = reinterpret_cast<const float2 *>(means2d)[idx];
            mx = m.x; my = m.y;
            t = tile_rect(mx, my, r, 16, tw, th, 0);
            cnt = (t.x1 - t.x0) * (t.y1 - t.y0);
            op = opacities[idx];
            ca = conics[idx * 3 + 0]; cb = conics[idx * 3 + 1]; cc = conics[idx * 3 + 2];
            if (EMIT) {
                const int64_t c = idx / N;
                key_hi_base = c << (32 + tile_bits);
                depth_bits = (int64_t)(uint32_t)__float_as_int(depths[idx]);
                start = (idx == 0) ? 0 : cum_live[idx - 1];
            }
        }
        if (!EMIT) tiles_per_gauss[idx] = cnt;
    }
    const bool wide = cnt >= kWideTiles;
    int live = 0;
    if (cnt > 0 && !wide) {
        for (int y = t.y0; y < t.y1; ++y)
            for (int x = t.x0; x < t.x1; ++x) {
                const unsigned m = quadrant_mask(mx, my, op, ca, cb, cc, (float)(x * 16), (float)(y * 16));
                if (m) {
                    if (EMIT) {
                        isect_ids[start + live] = key_hi_base | ((int64_t)(y * tw + x) << 32) | depth_bits;
                        payload[start + live] = (int32_t)((m << 28) | (uint32_t)idx);
                    }
                    ++live;
                }
            }
    }
    // wave-cooperative path for wide splats: 64 tiles per step, ballot-compacted
    unsigned long long wide_mask = __ballot(wide);
    while (wide_mask) {
        const int src = __ffsll((long long)wide_mask) - 1;
        wide_mask &= wide_mask - 1;
        const int x0 = __shfl(t.x0, src, 64), y0 = __shfl(t.y0, src, 64), x1 = __shfl(t.x1, src, 64);
        const int n = __shfl(cnt, src, 64);
        const float smx = __shfl(mx, src, 64), smy = __shfl(my, src, 64), sop = __shfl(op, src, 64);
        const float sca = __shfl(ca, src, 64), scb = __shfl(cb, src, 64), scc = __shfl(cc, src, 64);
        const int64_t kb = __shfl(key_hi_base, src, 64), db = __shfl(depth_bits, src, 64);
        const int64_t st = __shfl(start, src, 64);
        const uint32_t gid = (uint32_t)__shfl((long long)idx, src, 64);
        const int wdt = x1 - x0;
        int run = 0;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            unsigned m = 0u;
            int x = 0, y = 0;
            if (k < n) {
                y = y0 + k / wdt; x = x0 + k % wdt;
                m = quadrant_mask(smx, smy, sop, sca, scb, scc, (float)(x * 16), (float)(y * 16));
            }
            const unsigned long long b = __ballot(m != 0u);
            if (EMIT && m) {
                const int pos = run + __popcll(b & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                isect_ids[st + pos] = kb | ((int64_t)(y * tw + x) << 32) | db;
                payload[st + pos] = (int32_t)((m << 28) | gid);
            }
            run += __popcll(b);
        }
        if (lane == src) live = run;
    }
    if (!EMIT && idx < total) live_per_gauss[idx] = live;
}

// The same two passes with the (Gaussian, tile) pairs of a wave's 64 Gaussians FLATTENED over its
// lanes: a Gaussian covers 7 tiles on average but hundreds at the tail, and with one thread per
// Gaussian a wave lasted as long as its largest splat (47 + 51 us on BASELINE config #2, the mask
// arithmetic itself is ~5 us of the machine).  Per window of 64 consecutive pairs a lane finds its
// Gaussian by bisection over the wave's exclusive prefix of rectangle sizes (LDS), evaluates the
// quadrant mask of its own tile, and the live pairs of each Gaussian are ranked with one ballot.
struct FlatLds {
    int excl[64];       // exclusive prefix of the rectangle sizes
    float4 a[64];       // CullPrep: mx, my, b, tau
    float4 b[64];       // CullPrep: ha, hc, inv_a, inv_c
    float depth[64];    // EMIT: the sort key's low word
    int4 rect[64];      // x0, y0, width, count
    int run[64];        // live pairs of the Gaussian so far
    long long start[64];  // EMIT: first output position of the Gaussian
};

template <bool EMIT>
__global__ void __launch_bounds__(256)
isect_live_flat_kernel(int C, int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                       const float *__restrict__ depths, const float *__restrict__ conics,
                       const float *__restrict__ opacities, const int64_t *__restrict__ cum_live, int tw,
                       int th, int tile_bits, int32_t *__restrict__ tiles_per_gauss,
                       int32_t *__restrict__ live_per_gauss, int64_t *__restrict__ isect_ids,
                       int32_t *__restrict__ payload) {
    __shared__ FlatLds Ls[4];
    const int64_t total = (int64_t)C * N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    FlatLds &L = Ls[w];
    const int64_t idx0 = (int64_t)blockIdx.x * blockDim.x + w * 64;  // the wave's first Gaussian
    const int64_t idx = idx0 + lane;
    int cnt = 0;
    {
        TileRect t = {0, 0, 0, 0};
        float mx = 0.f, my = 0.f, op = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, db = 0.f;
        if (idx < total) {
            const int r = radii[idx];
            if (r > 0) {
                const float2 m = reinterpret_cast<const float2 *>(means2d)[idx];
                mx = m.x; my = m.y;
                t = tile_rect(mx, my, r, 16, tw, th, 0);
                cnt = (t.x1 - t.x0) * (t.y1 - t.y0);
                op = opacities[idx];
                ca = conics[idx * 3 + 0]; cb = conics[idx * 3 + 1]; cc = conics[idx * 3 + 2];
                if (EMIT) {
                    db = depths[idx];
                    L.start[lane] = (idx == 0) ? 0 : cum_live[idx - 1];
                }
            }
            if (!EMIT) tiles_per_gauss[idx] = cnt;
        }
        const CullPrep cp = cull_prepare(mx, my, op, ca, cb, cc);
        L.a[lane] = make_float4(cp.mx, cp.my, cp.b, cp.tau);
        L.b[lane] = make_float4(cp.ha, cp.hc, cp.inv_a, cp.inv_c);
        if (EMIT) L.depth[lane] = db;
        L.rect[lane] = make_int4(t.x0, t.y0, t.x1 - t.x0, cnt);
        L.run[lane] = 0;
    }
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    L.excl[lane] = inc - cnt;
    const int wave_total = __shfl(inc, 63, 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (int base = 0; base < wave_total; base += 64) {
        const int p = base + lane;
        const bool valid = p < wave_total;
        // owner = last lane whose exclusive prefix is <= p (empty rectangles share the prefix of the
        // next non-empty one, so "last" skips them)
        int lo = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const int mid = lo + step;
            if (L.excl[mid] <= p) lo = mid;
        }
        const int o = lo;
        const int4 rc = L.rect[o];
        const int first = L.excl[o];
        unsigned m = 0u;
        int x = 0, y = 0;
        if (valid) {
            const float4 A = L.a[o], B = L.b[o];
            const CullPrep cp = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};
            const int jj = p - first;
            y = rc.y + jj / rc.z; x = rc.x + jj - (jj / rc.z) * rc.z;
            m = quadrant_mask(cp, (float)(x * 16), (float)(y * 16));
        }
        const unsigned long long bal = __ballot(m != 0u);
        // lanes [s0, s1) of this window belong to the same Gaussian
        const int s0 = max(first - base, 0), s1 = min(max(first + rc.w - base, 0), 64);
        const unsigned long long upto = (lane == 0) ? 0ull : (~0ull >> (64 - lane));          // bits [0, lane)
        const unsigned long long from = (s0 == 0) ? ~0ull : (~0ull << s0);                     // bits [s0, 64)
        const unsigned long long to = (s1 >= 64) ? ~0ull : ((1ull << s1) - 1ull);              // bits [0, s1)
        const int before = L.run[o];
        if (EMIT && m) {
            const int64_t pos = L.start[o] + before + __popcll(bal & from & upto);
            const int64_t gidx = idx0 + o;
            const int64_t c = (C == 1) ? 0 : gidx / N;
            const int64_t dbits = (int64_t)(uint32_t)__float_as_int(L.depth[o]);
            isect_ids[pos] = (c << (32 + tile_bits)) | ((int64_t)(y * tw + x) << 32) | dbits;
            payload[pos] = (int32_t)((m << 28) | (uint32_t)gidx);
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && lane == s0) L.run[o] = before + __popcll(bal & from & to);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (!EMIT && idx < total) live_per_gauss[idx] = L.run[lane];
}

// ---- direct binning (row a-5/a-6/a-7 in one chain): the live pairs go straight into their tile's bucket ----
// The count -> scan -> emit -> tile_hist -> tile_scan -> tile_offsets -> tile_scatter chain writes every
// pair to a list in emission order only to read it back twice for the partition by tile.  Here the two
// enumeration passes do the partition themselves:
//   pass 1 (SCATTER = false): a workgroup of 16 waves enumerates the pairs of its 1024 Gaussians (flattened
//       over the lanes as above) and counts them per tile in LDS -> table[tile][workgroup]
//   tile_scan / tile_offsets (tilesort.hip): exclusive prefix over the workgroups per tile, isect_offsets
//   pass 2 (SCATTER = true): the same enumeration; every live pair claims the next slot of its tile's
//       bucket (LDS cursor, started at isect_offsets[tile] + table[workgroup][tile]) and drops its sort
//       word  bits(depth) << 32 | flatten id << 4 | quadrant mask  there
// followed by the in-tile sorts.  Four launches fewer, no emission-order arrays, no per-Gaussian scan.
// The claiming order inside a bucket is arbitrary; the in-tile sort on (depth, id) makes the result
// deterministic and bit-identical to the list chain's.
// Workgroup size: 1024 threads at N >= 256 k; smaller scenes take 512 / 256 so that the launch still has >= 256
// workgroups (config #3's 75 k Gaussians in chunks of 1024 would occupy 74 of the 256 CUs).
constexpr int kBinThreadsMax = 1024;

// per workgroup: its 1024 Gaussians' reach-test constants and tile rectangles.  The pairs of ALL of them are
// flattened over the 1024 threads (pair p of the workgroup -> its Gaussian by bisection over the exclusive
// prefix of the rectangle sizes), so every wave gets the same share however large single splats are.
template <int BT>
struct BinLds {
    int excl[BT];     // exclusive prefix of the rectangle sizes
    float4 a[BT];     // CullPrep: mx, my, b, tau
    float4 b[BT];     // CullPrep: ha, hc, inv_a, inv_c
    int4 rect[BT];    // x0, y0, width, count
    uint32_t depth[BT];
    int wave_sum[BT / 64];
};

// PROJ (count pass, one camera): the workgroup PROJECTS its Gaussians itself (project_math.h, with the activations
// and the binary-opacity write of fsgs_project_fwd_act) and writes the projection's outputs for the later passes:
// no projection launch, and the count pass reads 48 B of parameters instead of 28 B of projected attributes.
struct BinProjArgs {
    const float *means;
    const void *quats, *log_scales;  // fp32, or half mirrors when attr_half (common.h)
    const float *viewmat, *K;
    float *opac_logit;               // fp32 master (rewritten by the binary-opacity write)
    float *scales_out, *opac_out, *means2d, *depths, *conics;
    int32_t *radii;
    int width, height, binarise;
    float eps2d, near_plane, far_plane, radius_clip, binary_threshold;
    __half *opac_logit_h;            // attr_half: the mirror the opacity is READ from (and rewritten together with the master)
    int attr_half;
    // Occlusion cut (nullable; both binning passes of a frame get the same array): tile_zcut[t] = a depth beyond which
    // tile t's pairs are NOT binned at all.  The caller derived it from an earlier frame of the same view (the depth at
    // which every pixel of the tile had become opaque, plus a margin) and validates the frame afterwards
    // (fsgs_tile_zcut_update): the kept pairs are a depth PREFIX of the tile's full list, so a frame whose tiles all
    // saturate inside their prefix is bit-identical to the uncut one; any other frame is redone without the cut.
    const float *tile_zcut;
    int zcut_in_lds;                 // the workgroup keeps a copy of tile_zcut behind its tile slots (T more LDS words)
    int zcut_gw, zcut_gh;            // (zcut_in_lds) coarse grid of 4x4-tile blocks behind the copy: each block's largest cut
    // count pass only (nullable, [N] bytes): kept[n] <- 1 if Gaussian n has at least one binned (live, not cut) pair,
    // else 0.  A Gaussian without one is in no list: the frame needs neither its colours nor its packed record.
    uint8_t *kept;
    // count pass only (nullable): a word the first workgroup zeroes — the ticket of the scan launch that follows
    // (scan_rows_sh_pack_kernel: the last scan workgroup to arrive forms isect_offsets)
    int32_t *zero_word;
    // FUSED (count-free route, fsgs_project_bin_live_fill_sh_pack): fill_base[t] .. [t + 1] = tile t's bucket in
    // `buckets` (room from the view's previous frame), fill_cursor[t] = words claimed in it so far (zero on entry)
    const int32_t *fill_base;
    int32_t *fill_cursor;
};

template <bool SCATTER, bool MULTI, bool PROJ, bool HALF, int BT, bool FUSED = false>
__device__ __forceinline__ void
isect_live_bin_body(const int block, int C, int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                    const float *__restrict__ depths, const float *__restrict__ conics,
                    const float *__restrict__ opacities, int tw, int th, int T, int nb, int chunks,
                    int32_t *__restrict__ tiles_per_gauss, int32_t *__restrict__ table,
                    const int32_t *__restrict__ offsets, uint64_t *__restrict__ buckets, const BinProjArgs &pj,
                    int bucket_cap, int gpw = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bin_smem[];
    constexpr int kBinThreads = BT, kBinWaves = BT / 64;
    // gpw (count-free route, small models): Gaussians per workgroup when that is FEWER than its threads — the first gpw
    // threads project one Gaussian each, ALL threads share the flattened pairs.  A model of 70 k Gaussians with 23
    // rectangle pairs each (config #3) is 274 workgroups of 256 Gaussians: with 256 threads the chip held one wave per
    // SIMD and the two pair walks were 16 of the pass's 40 us; with 1024 threads the pass takes 23 us.  Fewer Gaussians
    // per workgroup would have doubled the (workgroup, tile) claims instead.  Measured: DESIGN.md section 5.1.
    const int gauss_per_wg = gpw > 0 ? gpw : kBinThreads;
    BinLds<BT> &L = *reinterpret_cast<BinLds<BT> *>(bin_smem);
    int *slots = reinterpret_cast<int *>(bin_smem + sizeof(BinLds<BT>));
    const int64_t total = (int64_t)C * N;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (PROJ && !SCATTER && block == 0 && tid == 0 && pj.zero_word) *pj.zero_word = 0;  // (the next launch's ticket)
    // `chunks` groups of 1024 Gaussians per workgroup, one after the other into the same tile slots: the table has
    // one row per WORKGROUP, so large N is given more chunks instead of more rows (N / 1024 rows of T counters
    // each would be 234 MB at 6 M Gaussians x 10 000 tiles)
    const int n_chunks = MULTI ? chunks : 1;  // (MULTI = false: the single-chunk code without the loop, 4 us faster)
    // Occlusion cuts: a copy of the frame's cuts behind the tile slots, and behind it the largest cut of every block of
    // 4x4 tiles — a Gaussian that lies behind the largest cut of every block its rectangle touches has no pair to bin
    // at all and leaves the pair loop (in a dense scene most of them: their pairs are nine in ten of all pairs)
    float *const zl = (pj.tile_zcut && pj.zcut_in_lds) ? reinterpret_cast<float *>(slots + T) : nullptr;
    float *const zg = zl ? zl + T : nullptr;
    if (zl) {
        for (int t = tid; t < T; t += kBinThreads) zl[t] = pj.tile_zcut[t];
        __syncthreads();
        for (int g = tid; g < pj.zcut_gw * pj.zcut_gh; g += kBinThreads) {
            const int gy = g / pj.zcut_gw, gx = g - gy * pj.zcut_gw;
            float zm = 0.f;
            for (int dy = 0; dy < 4; ++dy)
                for (int dx = 0; dx < 4; ++dx) {
                    const int ty = gy * 4 + dy, tx = gx * 4 + dx;
                    if (ty < th && tx < tw) zm = fmaxf(zm, zl[ty * tw + tx]);
                }
            zg[g] = zm;
        }
        __syncthreads();
    }
    for (int ch = 0; ch < n_chunks; ++ch) {
    const int64_t idx0 = ((int64_t)block * n_chunks + ch) * gauss_per_wg;  // the chunk's first Gaussian
    // (a thread beyond the workgroup's Gaussians projects nothing: idx = total makes every "idx < total" below false)
    const int64_t idx = tid < gauss_per_wg ? idx0 + tid : total;
    if (idx0 >= total) break;
    if (ch) __syncthreads();  // the previous chunk's constants are no longer read
    int cnt = 0;
    {
        TileRect t = {0, 0, 0, 0};
        float mx = 0.f, my = 0.f, op = 0.f, ca = 0.f, cb = 0.f, cc = 0.f;
        uint32_t db = 0u;
        if (PROJ && idx < total) {
            const int64_t n = idx;  // (one camera)
            float ls_in[3];
            ld_attr3(pj.log_scales, n, HALF ? 1 : 0, ls_in);
            const float s_act[3] = {expf(ls_in[0]), expf(ls_in[1]), expf(ls_in[2])};
            pj.scales_out[n * 3 + 0] = s_act[0]; pj.scales_out[n * 3 + 1] = s_act[1]; pj.scales_out[n * 3 + 2] = s_act[2];
            float ol = HALF ? __half2float(pj.opac_logit_h[n]) : pj.opac_logit[n];
            if (pj.binarise) {  // the binary-opacity write of get_outputs (dn_model.py:492-503), on the parameter itself
                ol = (ol >= pj.binary_threshold) ? 1.f : 0.f;
                if (!HALF || pj.opac_logit) pj.opac_logit[n] = ol;
                if (HALF) pj.opac_logit_h[n] = __float2half(ol);
            }
            const float o_act = 1.f / (1.f + expf(-ol));
            pj.opac_out[n] = o_act;
            const ProjOut po = project_one(pj.means[n * 3 + 0], pj.means[n * 3 + 1], pj.means[n * 3 + 2],
                                           ld_attr4(pj.quats, n, HALF ? 1 : 0), s_act, pj.viewmat, pj.K,
                                           pj.width, pj.height, pj.eps2d, pj.near_plane, pj.far_plane, pj.radius_clip);
            pj.radii[n] = po.radius;
            reinterpret_cast<float2 *>(pj.means2d)[n] = make_float2(po.u, po.v);
            pj.depths[n] = po.depth;
            pj.conics[n * 3 + 0] = po.ca; pj.conics[n * 3 + 1] = po.cb; pj.conics[n * 3 + 2] = po.cc;
            if (po.radius > 0) {
                mx = po.u; my = po.v;
                t = tile_rect(mx, my, po.radius, 16, tw, th, 0);
                cnt = (t.x1 - t.x0) * (t.y1 - t.y0);
                op = o_act; ca = po.ca; cb = po.cb; cc = po.cc;
                db = (uint32_t)__float_as_int(po.depth);
            }
            if (tiles_per_gauss) tiles_per_gauss[idx] = cnt;
            if (!SCATTER && pj.kept) pj.kept[idx] = 0;  // (raised below, after a barrier, by whichever thread bins a pair)
        } else if (idx < total) {
            const int r = radii[idx];
            if (r > 0) {
                const float2 m = reinterpret_cast<const float2 *>(means2d)[idx];
                mx = m.x; my = m.y;
                t = tile_rect(mx, my, r, 16, tw, th, 0);
                cnt = (t.x1 - t.x0) * (t.y1 - t.y0);
                op = opacities[idx];
                ca = conics[idx * 3 + 0]; cb = conics[idx * 3 + 1]; cc = conics[idx * 3 + 2];
                if (SCATTER) db = (uint32_t)__float_as_int(depths[idx]);
            }
            if (!SCATTER && tiles_per_gauss) tiles_per_gauss[idx] = cnt;
        }
        const CullPrep cp = cull_prepare(mx, my, op, ca, cb, cc);
        // Enumerate only the tiles the level set {sigma <= tau} can touch: its axis-aligned extent
        // (sqrt(2 tau cov_xx), sqrt(2 tau cov_yy)) plus one pixel of slack, inside gsplat's 3-sigma square.  Every
        // tile left out fails the exact reach test by more than that pixel, so the live pairs are the same; on
        // config #2 28 % of the square's tiles are never visited (elongated and faint Gaussians).
        if (cnt > 0) {
            if (cp.tau == -__builtin_huge_valf()) {
                t.x1 = t.x0; t.y1 = t.y0;
            } else if (cp.tau < 3.0e38f) {
                const float det = 4.f * cp.ha * cp.hc - cp.b * cp.b;  // (positive: cull_prepare checked it)
                const float k = 2.f * fmaxf(cp.tau, 0.f) / det;
                const float ex = sqrtf(k * 2.f * cp.hc) + 1.f, ey = sqrtf(k * 2.f * cp.ha) + 1.f;
                // tile j holds the pixel centres 16 j + 0.5 .. 16 j + 15.5
                t.x0 = max(t.x0, (int)ceilf((mx - ex - 15.5f) * (1.f / 16.f)));
                t.x1 = min(t.x1, (int)floorf((mx + ex - 0.5f) * (1.f / 16.f)) + 1);
                t.y0 = max(t.y0, (int)ceilf((my - ey - 15.5f) * (1.f / 16.f)));
                t.y1 = min(t.y1, (int)floorf((my + ey - 0.5f) * (1.f / 16.f)) + 1);
            }
            cnt = max(t.x1 - t.x0, 0) * max(t.y1 - t.y0, 0);
            if (zg && cnt > 0) {
                const int bx0 = t.x0 >> 2, bx1 = (t.x1 - 1) >> 2, by0 = t.y0 >> 2, by1 = (t.y1 - 1) >> 2;
                if (bx1 - bx0 < 3 && by1 - by0 < 3) {  // (larger rectangles keep their per-pair tests)
                    float zm = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
                            zm = fmaxf(zm, zg[min(by0 + dy, by1) * pj.zcut_gw + min(bx0 + dx, bx1)]);
                    if (__uint_as_float(db) > zm) cnt = 0;
                }
            }
        }
        L.a[tid] = make_float4(cp.mx, cp.my, cp.b, cp.tau);
        L.b[tid] = make_float4(cp.ha, cp.hc, cp.inv_a, cp.inv_c);
        L.rect[tid] = make_int4(t.x0, t.y0, max(t.x1 - t.x0, 1), cnt);
        L.depth[tid] = db;
    }
    if (ch == 0) {  // (after the chunk's own loads were issued: the slot rows come from memory in the fill pass)
        for (int t = tid; t < T; t += kBinThreads)
            slots[t] = SCATTER ? offsets[t] + table[(int64_t)block * T + t] : 0;
    }
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) L.wave_sum[w] = inc;
    __syncthreads();
    int wbase = 0, wg_total = 0;
#pragma unroll
    for (int k = 0; k < kBinWaves; ++k) {
        const int v = L.wave_sum[k];
        wbase += (k < w) ? v : 0;
        wg_total += v;
    }
    L.excl[tid] = wbase + inc - cnt;
    __syncthreads();  // prefix, constants and tile slots ready
    const int n_tiles = tw * th;
    // Thread t takes the CONTIGUOUS pairs [t * per, (t + 1) * per) of the workgroup's flattened list: one bisection
    // and one division for its first pair, then it walks — next tile of the rectangle, next non-empty Gaussian when
    // the rectangle is exhausted.  (First version: pair p = t + k * 1024, i.e. a 10-step LDS bisection and an integer
    // division for EVERY pair, ~70 of its ~300 instructions.)  Every thread gets the same number of pairs, however
    // large single splats are.
    const int per = (wg_total + kBinThreads - 1) / kBinThreads;
    int p = tid * per;
    const int p_end = min(p + per, wg_total);
    unsigned long long mask_lo = 0ull, mask_hi = 0ull;  // (FUSED) the reach masks of this thread's pairs, 4 bits each
    const int p_first = p;
    if (p < p_end) {
        // owner = last Gaussian whose exclusive prefix is <= p (empty rectangles share the prefix of the
        // next non-empty one, so "last" skips them)
        int o = 0;
#pragma unroll
        for (int step = kBinThreads / 2; step >= 1; step >>= 1) {
            const int mid = o + step;
            if (L.excl[mid] <= p) o = mid;
        }
        int4 rc = L.rect[o];
        float4 A = L.a[o], B = L.b[o];
        int jj = p - L.excl[o];
        int y = rc.y + jj / rc.z, x = rc.x + jj % rc.z;
        const float *zc = pj.tile_zcut ? (zl ? zl : pj.tile_zcut) : nullptr;
        float z_o = __uint_as_float(L.depth[o]);
        bool told = false;  // this thread has raised its current owner's kept flag
        for (; p < p_end; ++p) {
            const CullPrep cp = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};
            // (occlusion cut, one camera: a pair behind its tile's cut depth is not binned)
            const bool cut = zc && z_o > zc[y * tw + x];
            const unsigned m = cut ? 0u : quadrant_mask(cp, (float)(x * 16), (float)(y * 16));
            if (FUSED) {  // (the second walk reads the masks back: a 128-bit shift register, no indexed registers)
                mask_hi = (mask_hi << 4) | (mask_lo >> 60);
                mask_lo = (mask_lo << 4) | (unsigned long long)m;
            }
            if (m) {
                const int64_t gidx = idx0 + o;
                const int c = (C == 1) ? 0 : (int)(gidx / N);
                const int slot = atomicAdd(&slots[c * n_tiles + y * tw + x], 1);
                if (!SCATTER && PROJ && pj.kept && !told) { pj.kept[gidx] = 1; told = true; }
                // (slot >= bucket_cap only when the caller sized the buffers from an estimate that the frame
                // exceeded: the offsets were clamped to it by tile_offsets_kernel, the frame is redone)
                if (SCATTER && slot < bucket_cap)
                    buckets[slot] = ((uint64_t)L.depth[o] << 32) | ((uint64_t)(uint32_t)gidx << 4) | (uint64_t)m;
            }
            if (++jj == rc.w) {
                if (p + 1 < p_end) {  // (so a non-empty Gaussian follows)
                    do { ++o; } while (L.rect[o].w == 0);
                    rc = L.rect[o]; A = L.a[o]; B = L.b[o];
                    z_o = __uint_as_float(L.depth[o]);
                    told = false;
                    jj = 0; x = rc.x; y = rc.y;
                }
            } else if (++x == rc.x + rc.z) {
                x = rc.x; ++y;
            }
        }
    }
    if (FUSED) {
        // ---- count-free route: claim room in every touched tile's bucket, then walk the same pairs again and fill ----
        // (one returning atomic per (workgroup, touched tile) — ~400 per workgroup of 1024 Gaussians — instead of the
        // [workgroup][tile] table, its column scan and the second launch's re-enumeration)
        constexpr int kNoRoom = 0x40000000;
        __syncthreads();  // every pair of the workgroup is counted
        for (int t = tid; t < T; t += kBinThreads) {
            const int c = slots[t];
            int sl = kNoRoom;
            if (c > 0) {
                const int start = atomicAdd(&pj.fill_cursor[t], c);
                const int b0 = pj.fill_base[t];
                if (start + c <= pj.fill_base[t + 1] - b0) sl = b0 + start;  // (else: the frame is redone, see fsgs.h)
            }
            slots[t] = sl;
        }
        __syncthreads();
        p = p_first;
        if (p < p_end) {
            const int n_mine = p_end - p_first;
            const bool stored = n_mine <= 32;  // (else the masks did not fit: evaluated again)
            int o = 0;
#pragma unroll
            for (int step = kBinThreads / 2; step >= 1; step >>= 1) {
                const int mid = o + step;
                if (L.excl[mid] <= p) o = mid;
            }
            int4 rc = L.rect[o];
            float4 A = L.a[o], B = L.b[o];
            int jj = p - L.excl[o];
            int y = rc.y + jj / rc.z, x = rc.x + jj % rc.z;
            for (int k = 0; p < p_end; ++p, ++k) {
                unsigned m;
                if (stored) {
                    const int pos = 4 * (n_mine - 1 - k);
                    m = (unsigned)((pos < 64 ? (mask_lo >> pos) : (mask_hi >> (pos - 64))) & 0xFull);
                } else {
                    const CullPrep cp = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};
                    m = quadrant_mask(cp, (float)(x * 16), (float)(y * 16));
                }
                if (m) {
                    const int64_t gidx = idx0 + o;
                    const int slot = atomicAdd(&slots[y * tw + x], 1);
                    if (slot < bucket_cap)
                        buckets[slot] = ((uint64_t)L.depth[o] << 32) | ((uint64_t)(uint32_t)gidx << 4) | (uint64_t)m;
                }
                if (++jj == rc.w) {
                    if (p + 1 < p_end) {
                        do { ++o; } while (L.rect[o].w == 0);
                        rc = L.rect[o]; A = L.a[o]; B = L.b[o];
                        jj = 0; x = rc.x; y = rc.y;
                    }
                } else if (++x == rc.x + rc.z) {
                    x = rc.x; ++y;
                }
            }
        }
    }
    }  // chunks
    if (!SCATTER && !FUSED) {
        __syncthreads();
        for (int t = tid; t < T; t += kBinThreads) table[(int64_t)block * T + t] = slots[t];
    }
}

template <bool SCATTER, bool MULTI, bool PROJ, bool HALF = false, int BT = kBinThreadsMax>
__global__ void __launch_bounds__(BT)
isect_live_bin_kernel(int C, int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                      const float *__restrict__ depths, const float *__restrict__ conics,
                      const float *__restrict__ opacities, int tw, int th, int T, int nb, int chunks,
                      int32_t *__restrict__ tiles_per_gauss, int32_t *__restrict__ table,
                      const int32_t *__restrict__ offsets, uint64_t *__restrict__ buckets, BinProjArgs pj,
                      int bucket_cap = 0x7FFFFFFF) {
    isect_live_bin_body<SCATTER, MULTI, PROJ, HALF, BT>(blockIdx.x, C, N, means2d, radii, depths, conics, opacities, tw, th,
                                                        T, nb, chunks, tiles_per_gauss, table, offsets, buckets, pj,
                                                        bucket_cap);
}

// the count-free route's one binning launch (FUSED): project, count, claim, fill
template <bool HALF, int BT>
__global__ void __launch_bounds__(BT)
isect_live_bin_fill_kernel(int N, int tw, int th, int T, int nb, int32_t *__restrict__ tiles_per_gauss,
                           uint64_t *__restrict__ buckets, BinProjArgs pj, int bucket_cap, int gpw) {
    isect_live_bin_body<false, false, true, HALF, BT, true>(blockIdx.x, 1, N, nullptr, nullptr, nullptr, nullptr, nullptr, tw,
                                                            th, T, nb, 1, tiles_per_gauss, nullptr, nullptr, buckets, pj,
                                                            bucket_cap, gpw);
}

#ifndef FSGS_RIDE_ADAM_UNROLL
#define FSGS_RIDE_ADAM_UNROLL 2
#endif
constexpr int kRideAdamUnroll = FSGS_RIDE_ADAM_UNROLL;  // float4 runs per thread of the riding Adam blocks

// The projecting count pass with a deferred Adam step riding in its grid: nb workgroups project and count
// (vector-ALU and latency bound), the others stream the parameters, moments and gradients of a step that none of the
// count pass's inputs depend on (the SH features: bandwidth bound) — one launch instead of two back to back.
template <bool MULTI, bool HALF, int BT>
__global__ void __launch_bounds__(BT)
isect_count_adam_kernel(int N, int tw, int th, int T, int nb, int chunks, int32_t *__restrict__ tiles_per_gauss,
                        int32_t *__restrict__ table, BinProjArgs pj, AdamArgs adam) {
    // the count pass's workgroups are spread evenly through the grid (every stride-th block id), so that at any time
    // the CUs hold a mix of the two kinds of work: workgroups are dispatched in id order, and with all count blocks
    // first the memory system would idle until they retire (a CU has room for one 1024-thread workgroup of this kernel)
    const int b = blockIdx.x, stride = (nb + adam.n_blocks) / nb;
    const int slot = b / stride;
    if (b % stride != 0 || slot >= nb) {
        adam_body<kRideAdamUnroll>(adam, b - min(nb, (b + stride - 1) / stride), BT);
        return;
    }
    isect_live_bin_body<false, MULTI, true, HALF, BT>(slot, 1, N, nullptr, nullptr, nullptr, nullptr, nullptr, tw, th,
                                                      T, nb, chunks, tiles_per_gauss, table, nullptr, nullptr, pj,
                                                      0x7FFFFFFF);
}

__global__ void __launch_bounds__(256)
offset_encode_kernel(int64_t n_isects, const int64_t *__restrict__ ids, int n_tiles, int tile_bits,
                     int n_total, int32_t *__restrict__ offsets) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_isects) return;
    const int64_t hi = ids[idx] >> 32;
    const int64_t cur = (hi >> tile_bits) * n_tiles + (hi & ((1ll << tile_bits) - 1));
    if (idx == 0) {
        for (int64_t i = 0; i <= cur; ++i) offsets[i] = 0;
    } else {
        const int64_t hp = ids[idx - 1] >> 32;
        const int64_t prev = (hp >> tile_bits) * n_tiles + (hp & ((1ll << tile_bits) - 1));
        for (int64_t i = prev + 1; i <= cur; ++i) offsets[i] = (int32_t)idx;
    }
    if (idx == n_isects - 1) {
        for (int64_t i = cur + 1; i < n_total; ++i) offsets[i] = (int32_t)n_isects;
    }
}

inline int tile_bits_for(int n_tiles) {
    int b = 0;
    while ((1ll << b) <= n_tiles) ++b;  // floor(log2(n_tiles)) + 1
    return b < 1 ? 1 : b;
}

}  // namespace fsgs

using namespace fsgs;

extern "C" size_t fsgs_scan_scratch_bytes(int64_t n) {
    return (size_t)(scan_num_blocks(n) + 2) * sizeof(int64_t);  // + one spare word (rule_diff counter)
}

extern "C" int fsgs_isect_count(int C, int N, const float *means2d, const int32_t *radii,
                                int tile_size, int tile_width, int tile_height, int legacy,
                                int32_t *tiles_per_gauss, int64_t *cum_tiles, void *scratch,
                                size_t scratch_bytes, int64_t *n_isects_host,
                                int64_t *n_rule_diff_host, fsgs_stream_t stream) {
    if (C < 0 || N < 0 || tile_size < 1 || !n_isects_host) return FSGS_EINVAL;
    const int64_t total = (int64_t)C * N;
    *n_isects_host = 0;
    if (n_rule_diff_host) *n_rule_diff_host = 0;
    if (total == 0) return FSGS_OK;
    if (!means2d || !radii || !tiles_per_gauss || !cum_tiles) return FSGS_EINVAL;
    if (scratch_bytes < fsgs_scan_scratch_bytes(total) || !scratch) return FSGS_ESCRATCH;
    hipStream_t s = as_stream(stream);
    unsigned long long *diff_dev = nullptr;
    hipError_t e = hipSuccess;
    if (n_rule_diff_host) {
        diff_dev = reinterpret_cast<unsigned long long *>(scratch) + (scan_num_blocks(total) + 1);
        e = hipMemsetAsync(diff_dev, 0, sizeof(unsigned long long), s);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
    }
    hipLaunchKernelGGL(isect_count_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, total, means2d,
                       radii, tile_size, tile_width, tile_height, legacy, tiles_per_gauss, diff_dev);
    int rc = device_scan<int32_t, true>(total, tiles_per_gauss, cum_tiles, scratch, scratch_bytes, s);
    if (rc != FSGS_OK) return rc;
    e = hipMemcpyAsync(n_isects_host, cum_tiles + (total - 1), sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && n_rule_diff_host)
        e = hipMemcpyAsync(n_rule_diff_host, diff_dev, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return FSGS_ELAUNCH;
    }
    return FSGS_OK;
}

extern "C" int fsgs_isect_emit(int C, int N, const float *means2d, const int32_t *radii,
                               const float *depths, const int64_t *cum_tiles, int tile_size,
                               int tile_width, int tile_height, int legacy, int64_t *isect_ids,
                               int32_t *flatten_ids, fsgs_stream_t stream) {
    if (C < 0 || N < 0 || tile_size < 1) return FSGS_EINVAL;
    const int64_t total = (int64_t)C * N;
    if (total == 0) return FSGS_OK;
    if (!means2d || !radii || !depths || !cum_tiles || !isect_ids || !flatten_ids) return FSGS_EINVAL;
    const int tb = tile_bits_for(tile_width * tile_height);
    hipLaunchKernelGGL(isect_emit_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), C,
                       N, means2d, radii, depths, cum_tiles, tile_size, tile_width, tile_height, legacy,
                       tb, isect_ids, flatten_ids);
    return check_launch();
}

extern "C" int fsgs_isect_offset_encode(int64_t n_isects, const int64_t *isect_ids_sorted, int C,
                                        int n_tiles, int tile_bits, int32_t *offsets,
                                        fsgs_stream_t stream) {
    if (n_isects < 0 || C < 0 || n_tiles < 0 || !offsets) return FSGS_EINVAL;
    const int n_total = C * n_tiles;
    hipStream_t s = as_stream(stream);
    if (n_isects == 0) {
        hipError_t e = hipMemsetAsync(offsets, 0, (size_t)n_total * sizeof(int32_t), s);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
        return FSGS_OK;
    }
    if (!isect_ids_sorted) return FSGS_EINVAL;
    hipLaunchKernelGGL(offset_encode_kernel, dim3(ceil_div(n_isects, 256)), dim3(256), 0, s, n_isects,
                       isect_ids_sorted, n_tiles, tile_bits, n_total, offsets);
    return check_launch();
}


// Live emission, pass 1: tiles_per_gauss (gsplat's rectangle count), live_per_gauss and its inclusive
// scan cum_live; synchronises and returns the number of LIVE intersections.  tile_size is 16.
extern "C" int fsgs_isect_count_live(int C, int N, const float *means2d, const int32_t *radii,
                                     const float *conics, const float *opacities, int tile_width,
                                     int tile_height, int32_t *tiles_per_gauss, int32_t *live_per_gauss,
                                     int64_t *cum_live, void *scratch, size_t scratch_bytes,
                                     int64_t *n_live_host, fsgs_stream_t stream) {
    if (C < 0 || N < 0) return FSGS_EINVAL;
    const int64_t total = (int64_t)C * N;
    if (n_live_host) *n_live_host = 0;
    if (total == 0) return FSGS_OK;
    if (total >= (1ll << 28)) return FSGS_EINVAL;  // the payload keeps 28 bits for the flatten id
    if (!means2d || !radii || !conics || !opacities || !tiles_per_gauss || !live_per_gauss || !cum_live)
        return FSGS_EINVAL;
    if (scratch_bytes < fsgs_scan_scratch_bytes(total) || !scratch) return FSGS_ESCRATCH;
    hipStream_t s = as_stream(stream);
    constexpr bool per_thread = FSGS_ISECT_PER_THREAD != 0;  // (build macro: the library reads no environment)
    if (per_thread)
        hipLaunchKernelGGL((isect_live_kernel<false>), dim3(ceil_div(total, 256)), dim3(256), 0, s, C, N, means2d,
                           radii, nullptr, conics, opacities, nullptr, tile_width, tile_height, 0, tiles_per_gauss,
                           live_per_gauss, nullptr, nullptr);
    else
        hipLaunchKernelGGL((isect_live_flat_kernel<false>), dim3(ceil_div(total, 256)), dim3(256), 0, s, C, N,
                           means2d, radii, nullptr, conics, opacities, nullptr, tile_width, tile_height, 0,
                           tiles_per_gauss, live_per_gauss, nullptr, nullptr);
    int rc = device_scan<int32_t, true>(total, live_per_gauss, cum_live, scratch, scratch_bytes, s);
    if (rc != FSGS_OK) return rc;
    if (!n_live_host) return check_launch();  // the caller reads cum_live[total-1] back itself (asynchronously)
    hipError_t e = hipMemcpyAsync(n_live_host, cum_live + (total - 1), sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
    return FSGS_OK;
}

// Pass 2: isect_ids[M_live] and payload[M_live] = quadrant mask << 28 | flatten id.
extern "C" int fsgs_isect_emit_live(int C, int N, const float *means2d, const int32_t *radii,
                                    const float *depths, const float *conics, const float *opacities,
                                    const int64_t *cum_live, int tile_width, int tile_height,
                                    int64_t *isect_ids, int32_t *payload, fsgs_stream_t stream) {
    if (C < 0 || N < 0) return FSGS_EINVAL;
    const int64_t total = (int64_t)C * N;
    if (total == 0) return FSGS_OK;
    if (!means2d || !radii || !depths || !conics || !opacities || !cum_live || !isect_ids || !payload)
        return FSGS_EINVAL;
    const int tb = tile_bits_for(tile_width * tile_height);
    constexpr bool per_thread = FSGS_ISECT_PER_THREAD != 0;  // (build macro: the library reads no environment)
    if (per_thread)
        hipLaunchKernelGGL((isect_live_kernel<true>), dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), C,
                           N, means2d, radii, depths, conics, opacities, cum_live, tile_width, tile_height, tb,
                           nullptr, nullptr, isect_ids, payload);
    else
        hipLaunchKernelGGL((isect_live_flat_kernel<true>), dim3(ceil_div(total, 256)), dim3(256), 0,
                           as_stream(stream), C, N, means2d, radii, depths, conics, opacities, cum_live, tile_width,
                           tile_height, tb, nullptr, nullptr, isect_ids, payload);
    return check_launch();
}

// ---- direct binning: C entry points -----------------------------------------------------------------------
namespace fsgs {
int launch_tile_scan_rows_offsets(int T, int nb, int32_t *table, int32_t *totals, int32_t *isect_offsets,
                                  int32_t *total_mapped, hipStream_t s);
int launch_tile_offsets(int T, const int32_t *totals, int32_t *isect_offsets, int32_t *total_mapped, hipStream_t s);
int launch_scan_rows_sh_pack(int T, int nb, int32_t *table, int32_t *totals, const ShPackRider &r, hipStream_t s,
                             int32_t *ticket, int32_t *isect_offsets, int32_t *total_mapped, const BucketBook &bk);
int launch_tile_sort_tiers(int T, int n_tiles, int tile_bits, const int32_t *isect_offsets, uint64_t *buckets,
                           int32_t *payload_sorted, int64_t *isect_ids_sorted, hipStream_t s, int32_t *long_flag = nullptr,
                           int rel_gate = 0, const int32_t *src_offsets = nullptr, const int32_t *tile_order = nullptr);
int launch_split_sort(int T, int n_tiles, int tile_bits, const int32_t *isect_offsets, int64_t n_live,
                      const uint64_t *buckets, uint64_t *buckets2, int32_t *scratch, int32_t *payload_sorted,
                      hipStream_t s);
int64_t split_scratch_ints(int T, int64_t n_live);
}

// Workgroup size of the binning passes: the largest of 1024 / 512 / 256 threads that still gives FSGS_BIN_WGS_MIN
// workgroups — TWO per CU (round 5; rounds 1-4: one).  The passes are chains of dependent phases (project, scan, enumerate,
// claim with returning atomics, store) with barriers in between: a second workgroup on the CU fills the first one's waits.
// Config #2 (300 k Gaussians: 586 workgroups of 512 instead of 293 of 1024): GPU step 0.515-0.518 -> 0.506-0.507 ms;
// 1172 workgroups of 256: 0.530 (every workgroup claims its own run in every tile it touches).
#ifndef FSGS_BIN_WGS_MIN
#define FSGS_BIN_WGS_MIN 512
#endif
static inline int bin_threads(int64_t total) {
    return total >= (int64_t)FSGS_BIN_WGS_MIN * 1024 ? 1024 : (total >= (int64_t)FSGS_BIN_WGS_MIN * 512 ? 512 : 256);
}
// chunks of 1024 Gaussians per workgroup: as many as keep the table at ~512 rows
static inline int bin_chunks(int64_t total) {
    const int64_t c = (total + (int64_t)kBinThreadsMax * 512 - 1) / ((int64_t)kBinThreadsMax * 512);
    return c < 1 ? 1 : (int)c;
}
static inline int64_t bin_blocks(int64_t total) {
    const int64_t per = (int64_t)bin_threads(total) * bin_chunks(total);
    return (total + per - 1) / per;
}
static inline size_t bin_lds_bytes(int T, int64_t total) {
    const int bt = bin_threads(total);
    const size_t fixed = bt == 1024 ? sizeof(BinLds<1024>) : (bt == 512 ? sizeof(BinLds<512>) : sizeof(BinLds<256>));
    return fixed + (size_t)T * sizeof(int);
}

// tiles (cameras x tiles) the direct path can take: the per-workgroup tile slots live in LDS (160 KB per CU)
extern "C" int fsgs_bin_live_max_tiles(void) { return 24576; }

// table scratch: [table i32: blocks x T] [totals i32: T]
extern "C" size_t fsgs_bin_live_table_bytes(int C, int N, int tile_width, int tile_height) {
    const int64_t T = (int64_t)C * tile_width * tile_height;
    const int64_t nb = bin_blocks((int64_t)C * N > 0 ? (int64_t)C * N : 1);
    return (size_t)(T * nb + T) * sizeof(int32_t) + 64;
}

// one launch of isect_live_bin_kernel<SC, multi?, PJ, HF, threads(total)>: dynamic LDS limit raised on first need
template <bool SC, bool MU, bool PJ, bool HF, int BT, typename... Args>
static int bin_launch_one(size_t extra_lds, int T, int64_t total, int nb, hipStream_t s, Args... args) {
    const size_t need = bin_lds_bytes(T, total) + extra_lds;
    auto kernel = &isect_live_bin_kernel<SC, MU, PJ, HF, BT>;
    if (const int rc = ensure_dynamic_lds<&isect_live_bin_kernel<SC, MU, PJ, HF, BT>>(need)) return rc;
    hipLaunchKernelGGL(kernel, dim3(nb), dim3(BT), need, s, args...);
    return check_launch();
}
template <bool SC, bool PJ, bool HF, typename... Args>
static int bin_launch(size_t extra_lds, int T, int64_t total, int nb, hipStream_t s, Args... args) {
    switch (bin_threads(total)) {
    case 1024:
        if (bin_chunks(total) > 1) return bin_launch_one<SC, true, PJ, HF, 1024>(extra_lds, T, total, nb, s, args...);
        return bin_launch_one<SC, false, PJ, HF, 1024>(extra_lds, T, total, nb, s, args...);
    case 512: return bin_launch_one<SC, false, PJ, HF, 512>(extra_lds, T, total, nb, s, args...);
    default: return bin_launch_one<SC, false, PJ, HF, 256>(extra_lds, T, total, nb, s, args...);
    }
}
// the tile cuts of a frame in LDS behind the tile slots, where both fit (T more words); else they are read from memory
static inline void bin_place_zcut(BinProjArgs &pj, const float *tile_zcut, int tw, int th, int T, int64_t total,
                                  size_t &extra_lds) {
    pj.tile_zcut = tile_zcut;
    pj.zcut_in_lds = 0;
    pj.zcut_gw = (tw + 3) / 4;
    pj.zcut_gh = (th + 3) / 4;
    extra_lds = 0;
    const size_t need = ((size_t)T + (size_t)pj.zcut_gw * pj.zcut_gh) * sizeof(float);
    if (tile_zcut && bin_lds_bytes(T, total) + need <= 150 * 1024) {
        pj.zcut_in_lds = 1;
        extra_lds = need;
    }
}

// the projecting count pass + a riding Adam step (isect_count_adam_kernel)
template <bool MU, bool HF, int BT>
static int count_adam_launch_one(size_t extra_lds, int T, int64_t total, int nb, hipStream_t s, int N, int tw, int th, int chunks,
                                 int32_t *tiles_per_gauss, int32_t *table, const BinProjArgs &pj, const AdamArgs &adam) {
    const size_t need = bin_lds_bytes(T, total) + extra_lds;
    auto kernel = &isect_count_adam_kernel<MU, HF, BT>;
    if (const int rc = ensure_dynamic_lds<&isect_count_adam_kernel<MU, HF, BT>>(need)) return rc;
    hipLaunchKernelGGL(kernel, dim3(nb + adam.n_blocks), dim3(BT), need, s, N, tw, th, T, nb, chunks, tiles_per_gauss,
                       table, pj, adam);
    return check_launch();
}
template <bool HF>
static int count_adam_launch(size_t extra_lds, int T, int64_t total, int nb, hipStream_t s, int N, int tw, int th,
                             int32_t *tiles_per_gauss, int32_t *table, const BinProjArgs &pj, const AdamArgs &adam) {
    const int chunks = bin_chunks(total);
    switch (bin_threads(total)) {
    case 1024:
        if (chunks > 1) return count_adam_launch_one<true, HF, 1024>(extra_lds, T, total, nb, s, N, tw, th, chunks, tiles_per_gauss, table, pj, adam);
        return count_adam_launch_one<false, HF, 1024>(extra_lds, T, total, nb, s, N, tw, th, chunks, tiles_per_gauss, table, pj, adam);
    case 512: return count_adam_launch_one<false, HF, 512>(extra_lds, T, total, nb, s, N, tw, th, chunks, tiles_per_gauss, table, pj, adam);
    default: return count_adam_launch_one<false, HF, 256>(extra_lds, T, total, nb, s, N, tw, th, chunks, tiles_per_gauss, table, pj, adam);
    }
}

// Pass 1 + the per-tile scans: tiles_per_gauss (gsplat's count, nullable), isect_offsets[T + 1] with
// isect_offsets[T] = the number of live pairs (read it back to size pass 2's buffers).
extern "C" int fsgs_bin_live_count(int C, int N, const float *means2d, const int32_t *radii, const float *conics,
                                   const float *opacities, int tile_width, int tile_height,
                                   int32_t *tiles_per_gauss, int32_t *isect_offsets, void *table_scratch,
                                   size_t table_bytes, int32_t *n_live_mapped, fsgs_stream_t stream) {
    if (C < 1 || N < 0 || tile_width < 1 || tile_height < 1 || !isect_offsets) return FSGS_EINVAL;
    const int64_t T64 = (int64_t)C * tile_width * tile_height;
    if (T64 > fsgs_bin_live_max_tiles()) return FSGS_EINVAL;
    const int T = (int)T64;
    hipStream_t s = as_stream(stream);
    const int64_t total = (int64_t)C * N;
    if (total == 0) {
        hipError_t e = hipMemsetAsync(isect_offsets, 0, (size_t)(T + 1) * sizeof(int32_t), s);
        if (e == hipSuccess && n_live_mapped) {
            static const int32_t kZeroDone[2] = {0, 1};
            e = hipMemcpyAsync(n_live_mapped, kZeroDone, sizeof(kZeroDone), hipMemcpyHostToHost, s);
        }
        if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
        return FSGS_OK;
    }
    if (!means2d || !radii || !conics || !opacities || !table_scratch) return FSGS_EINVAL;
    if (table_bytes < fsgs_bin_live_table_bytes(C, N, tile_width, tile_height)) return FSGS_ESCRATCH;
    const int nb = (int)bin_blocks(total);
    int32_t *table = reinterpret_cast<int32_t *>(table_scratch);
    int32_t *totals = table + (size_t)T * nb;
    int rc = bin_launch<false, false, false>((size_t)0, T, total, nb, s, C, N, means2d, radii, (const float *)nullptr, conics,
                                             opacities, tile_width, tile_height, T, nb, bin_chunks(total),
                                             tiles_per_gauss, table, (const int32_t *)nullptr, (uint64_t *)nullptr,
                                             BinProjArgs{}, 0x7FFFFFFF);
    if (rc != FSGS_OK) return rc;
    return launch_tile_scan_rows_offsets(T, nb, table, totals, isect_offsets, n_live_mapped, s);
}

// fsgs_project_fwd_act (one camera) + fsgs_bin_live_count in one launch chain: the count pass projects the
// Gaussians itself.  Outputs of both: scales_out, opac_out, radii, means2d, depths, conics [N..], tiles_per_gauss,
// isect_offsets[T + 1]; opac_logit is rewritten when binarise != 0.
static int project_bin_live_count_impl(int N, const float *means, const void *quats, const void *log_scales,
                                       float *opac_logit, void *opac_logit_h, int attr_half, int binarise,
                                       float binary_threshold,
                                           const float *viewmat, const float *K, int width, int height, float eps2d,
                                           float near_plane, float far_plane, float radius_clip, float *scales_out,
                                           float *opac_out, int32_t *radii, float *means2d, float *depths,
                                           float *conics, int tile_width, int tile_height, int32_t *tiles_per_gauss,
                                           int32_t *isect_offsets, void *table_scratch, size_t table_bytes,
                                           int32_t *n_live_mapped, fsgs_stream_t stream,
                                           const ShPackRider *rider = nullptr, const fsgs_adam_groups *adam = nullptr,
                                           const float *tile_zcut = nullptr, uint8_t *kept = nullptr,
                                           int32_t *tile_order = nullptr) {
    if (N < 0 || tile_width < 1 || tile_height < 1 || !isect_offsets) return FSGS_EINVAL;
    const int64_t T64 = (int64_t)tile_width * tile_height;
    if (T64 > fsgs_bin_live_max_tiles()) return FSGS_EINVAL;
    const int T = (int)T64;
    hipStream_t s = as_stream(stream);
    if (rider && (rider->degree < 0 || rider->degree > 3 || !rider->campos || !rider->c2w || (N > 0 &&
        (!rider->features_dc || !rider->features_rest || !rider->packed || !rider->normals_world))))
        return FSGS_EINVAL;
    if (N == 0) {
        hipError_t e = hipMemsetAsync(isect_offsets, 0, (size_t)(T + 1) * sizeof(int32_t), s);
        if (e == hipSuccess && n_live_mapped) {
            static const int32_t kZeroDone[2] = {0, 1};
            e = hipMemcpyAsync(n_live_mapped, kZeroDone, sizeof(kZeroDone), hipMemcpyHostToHost, s);
        }
        if (e == hipSuccess && rider && rider->zero_cells && rider->n_zero > 0)
            e = hipMemsetAsync(rider->zero_cells, 0, (size_t)rider->n_zero * sizeof(float), s);
        if (e == hipSuccess && adam)  // (nothing to ride in: the step as a launch of its own)
            return fsgs_adam_step_h16(adam->n_groups, adam->params, adam->grads, adam->exp_avg, adam->exp_avg_sq,
                                      adam->half_mirror, adam->numel, adam->lr, adam->step, adam->beta1, adam->beta2,
                                      adam->eps, stream);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
        return FSGS_OK;
    }
    if (!means || !quats || !log_scales || !viewmat || !K || !scales_out || !opac_out || !radii ||
        !means2d || !depths || !conics || !table_scratch)
        return FSGS_EINVAL;
    if (attr_half ? !opac_logit_h : !opac_logit) return FSGS_EINVAL;
    if (table_bytes < fsgs_bin_live_table_bytes(1, N, tile_width, tile_height)) return FSGS_ESCRATCH;
    const int64_t total = N;
    const int nb = (int)bin_blocks(total);
    int32_t *table = reinterpret_cast<int32_t *>(table_scratch);
    int32_t *totals = table + (size_t)T * nb;
    int rc;
    BinProjArgs pj = {means, quats, log_scales, viewmat, K, opac_logit, scales_out, opac_out, means2d, depths,
                      conics, radii, width, height, binarise, eps2d, near_plane, far_plane, radius_clip,
                      binary_threshold, reinterpret_cast<__half *>(opac_logit_h), attr_half, nullptr, 0, 0, 0, nullptr,
                      nullptr, nullptr, nullptr};
    size_t extra_lds = 0;
    bin_place_zcut(pj, tile_zcut, tile_width, tile_height, T, total, extra_lds);
    pj.kept = kept;
    int32_t *ticket = rider ? totals + T : nullptr;  // (inside table_scratch's 64 spare bytes)
    pj.zero_word = ticket;
    ShPackRider rider_kept;
    if (rider && kept) { rider_kept = *rider; rider_kept.kept = kept; rider = &rider_kept; }
#define FSGS_BIN_PCOUNT(HF)                                                                                         \
    bin_launch<false, true, HF>(extra_lds, T, total, nb, s, 1, N, (const float *)nullptr, (const int32_t *)nullptr,   \
                                (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, tile_width,  \
                                tile_height, T, nb, bin_chunks(total), tiles_per_gauss, table,                        \
                                (const int32_t *)nullptr, (uint64_t *)nullptr, pj, 0x7FFFFFFF)
    AdamArgs aa;
    aa.n_blocks = 0;
    if (adam) {  // a deferred Adam step rides in the count pass's grid
        rc = build_adam_args(aa, bin_threads(total) * kRideAdamUnroll, adam->n_groups, adam->params, adam->grads, adam->exp_avg,
                             adam->exp_avg_sq, adam->half_mirror, adam->numel, adam->lr, adam->step, adam->beta1,
                             adam->beta2, adam->eps);
        if (rc != FSGS_OK) return rc;
    }
    if (aa.n_blocks > 0)
        rc = attr_half ? count_adam_launch<true>(extra_lds, T, total, nb, s, N, tile_width, tile_height, tiles_per_gauss, table, pj, aa)
                       : count_adam_launch<false>(extra_lds, T, total, nb, s, N, tile_width, tile_height, tiles_per_gauss, table, pj, aa);
    else
        rc = attr_half ? FSGS_BIN_PCOUNT(true) : FSGS_BIN_PCOUNT(false);
#undef FSGS_BIN_PCOUNT
    if (rc != FSGS_OK) return rc;
    if (rider)  // the table scan, the SH forward + packing AND the offsets in one launch (sh.hip)
        return launch_scan_rows_sh_pack(T, nb, table, totals, *rider, s, ticket, isect_offsets, n_live_mapped,
                                        BucketBook{nullptr, nullptr, nullptr, 0.f, 0, tile_order});
    return launch_tile_scan_rows_offsets(T, nb, table, totals, isect_offsets, n_live_mapped, s);
}

// fsgs_project_bin_live_count followed by fsgs_sh_fwd_pack (K = 16 stored coefficients, split features), with the
// SH forward riding in the launch of the count pass's table scan: same outputs as the two calls.
extern "C" int fsgs_project_bin_live_count_sh_pack(
    int N, const float *means, const float *quats, const float *log_scales, float *opac_logit, int binarise,
    float binary_threshold, const float *viewmat, const float *K, int width, int height, float eps2d, float near_plane,
    float far_plane, float radius_clip, float *scales_out, float *opac_out, int32_t *radii, float *means2d,
    float *depths, float *conics, int tile_width, int tile_height, int32_t *tiles_per_gauss, int32_t *isect_offsets,
    void *table_scratch, size_t table_bytes, int32_t *n_live_mapped, int degree, const float *campos,
    const float *features_dc, const float *features_rest, const float *c2w, float *packed, float *normals_world,
    float *zero_cells, int n_zero, const fsgs_adam_groups *adam, const float *tile_zcut, uint8_t *kept,
    int32_t *tile_order, fsgs_stream_t stream) {
    const ShPackRider r = {N, degree, 0, means, campos, features_dc, features_rest, radii, depths, means2d, conics,
                           opac_out, quats, log_scales, c2w, packed, normals_world, zero_cells, n_zero, nullptr};
    return project_bin_live_count_impl(N, means, quats, log_scales, opac_logit, nullptr, 0, binarise, binary_threshold,
                                       viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip, scales_out,
                                       opac_out, radii, means2d, depths, conics, tile_width, tile_height, tiles_per_gauss,
                                       isect_offsets, table_scratch, table_bytes, n_live_mapped, stream,
                                       packed ? &r : nullptr, adam, tile_zcut, kept, packed ? tile_order : nullptr);
}

extern "C" int fsgs_project_bin_live_count_sh_pack_h16(
    int N, const float *means, const void *quats_h, const void *log_scales_h, void *opac_logit_h,
    float *opac_logit_master, int binarise, float binary_threshold, const float *viewmat, const float *K, int width,
    int height, float eps2d, float near_plane, float far_plane, float radius_clip, float *scales_out, float *opac_out,
    int32_t *radii, float *means2d, float *depths, float *conics, int tile_width, int tile_height,
    int32_t *tiles_per_gauss, int32_t *isect_offsets, void *table_scratch, size_t table_bytes, int32_t *n_live_mapped,
    int degree, const float *campos, const void *features_dc_h, const void *features_rest_h, const float *c2w,
    float *packed, float *normals_world, float *zero_cells, int n_zero, const fsgs_adam_groups *adam,
    const float *tile_zcut, uint8_t *kept, int32_t *tile_order, fsgs_stream_t stream) {
    const ShPackRider r = {N, degree, 1, means, campos, features_dc_h, features_rest_h, radii, depths, means2d, conics,
                           opac_out, quats_h, log_scales_h, c2w, packed, normals_world, zero_cells, n_zero, nullptr};
    return project_bin_live_count_impl(N, means, quats_h, log_scales_h, opac_logit_master, opac_logit_h, 1, binarise,
                                       binary_threshold, viewmat, K, width, height, eps2d, near_plane, far_plane,
                                       radius_clip, scales_out, opac_out, radii, means2d, depths, conics, tile_width,
                                       tile_height, tiles_per_gauss, isect_offsets, table_scratch, table_bytes,
                                       n_live_mapped, stream, packed ? &r : nullptr, adam, tile_zcut, kept,
                                       packed ? tile_order : nullptr);
}

extern "C" int fsgs_project_bin_live_count(int N, const float *means, const float *quats, const float *log_scales,
                                           float *opac_logit, int binarise, float binary_threshold,
                                           const float *viewmat, const float *K, int width, int height, float eps2d,
                                           float near_plane, float far_plane, float radius_clip, float *scales_out,
                                           float *opac_out, int32_t *radii, float *means2d, float *depths,
                                           float *conics, int tile_width, int tile_height, int32_t *tiles_per_gauss,
                                           int32_t *isect_offsets, void *table_scratch, size_t table_bytes,
                                           int32_t *n_live_mapped, fsgs_stream_t stream) {
    return project_bin_live_count_impl(N, means, quats, log_scales, opac_logit, nullptr, 0, binarise, binary_threshold,
                                       viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip, scales_out,
                                       opac_out, radii, means2d, depths, conics, tile_width, tile_height, tiles_per_gauss,
                                       isect_offsets, table_scratch, table_bytes, n_live_mapped, stream);
}

// Half attribute storage (BASELINE config #5): quats / log_scales / opac_logit_h are IEEE-half mirrors of the fp32
// masters; the binary-opacity write goes to the mirror AND to the master (opac_logit_master, nullable).
extern "C" int fsgs_project_bin_live_count_h16(int N, const float *means, const void *quats_h, const void *log_scales_h,
                                               void *opac_logit_h, float *opac_logit_master, int binarise,
                                               float binary_threshold, const float *viewmat, const float *K, int width,
                                               int height, float eps2d, float near_plane, float far_plane,
                                               float radius_clip, float *scales_out, float *opac_out, int32_t *radii,
                                               float *means2d, float *depths, float *conics, int tile_width,
                                               int tile_height, int32_t *tiles_per_gauss, int32_t *isect_offsets,
                                               void *table_scratch, size_t table_bytes, int32_t *n_live_mapped,
                                               fsgs_stream_t stream) {
    return project_bin_live_count_impl(N, means, quats_h, log_scales_h, opac_logit_master, opac_logit_h, 1, binarise,
                                       binary_threshold, viewmat, K, width, height, eps2d, near_plane, far_plane,
                                       radius_clip, scales_out, opac_out, radii, means2d, depths, conics, tile_width,
                                       tile_height, tiles_per_gauss, isect_offsets, table_scratch, table_bytes,
                                       n_live_mapped, stream);
}

// Pass 2 + the in-tile sorts: payload_sorted[n_live] = quadrant mask << 28 | flatten id in (tile, depth, id)
// order.  `buckets` = n_live 64-bit words of scratch; table_scratch / isect_offsets as left by pass 1.
static int bin_live_emit_impl(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                              const float *conics, const float *opacities, int tile_width, int tile_height,
                              const int32_t *isect_offsets, const void *table_scratch, int64_t n_live,
                              void *buckets, void *buckets2, void *split_scratch, int32_t *payload_sorted,
                              fsgs_stream_t stream, const float *tile_zcut, int32_t *long_flag = nullptr,
                              int rel_gate = 0);

extern "C" size_t fsgs_bin_live_split_scratch_bytes(int C, int tile_width, int tile_height, int64_t n_live) {
    return (size_t)split_scratch_ints(C * tile_width * tile_height, n_live > 0 ? n_live : 0) * sizeof(int32_t) + 64;
}

// fsgs_bin_live_emit for buckets of any size: tiles with more than 1024 live pairs are split into depth slabs of
// ~512 before the LDS sorts (tilesort.hip), so dense scenes need neither the radix sort nor its lists.
// buckets, buckets2: n_live 64-bit words each; split_scratch: fsgs_bin_live_split_scratch_bytes().
extern "C" int fsgs_bin_live_emit_split(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                                        const float *conics, const float *opacities, int tile_width, int tile_height,
                                        const int32_t *isect_offsets, const void *table_scratch, int64_t n_live,
                                        void *buckets, void *buckets2, void *split_scratch, size_t split_bytes,
                                        int32_t *payload_sorted, const float *tile_zcut, fsgs_stream_t stream) {
    if (n_live > 0 && (!buckets2 || !split_scratch)) return FSGS_EINVAL;
    if (n_live > 0 && split_bytes < fsgs_bin_live_split_scratch_bytes(C, tile_width, tile_height, n_live)) return FSGS_ESCRATCH;
    return bin_live_emit_impl(C, N, means2d, radii, depths, conics, opacities, tile_width, tile_height, isect_offsets,
                              table_scratch, n_live, buckets, buckets2, split_scratch, payload_sorted, stream, tile_zcut);
}

extern "C" int fsgs_bin_live_emit(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                                  const float *conics, const float *opacities, int tile_width, int tile_height,
                                  const int32_t *isect_offsets, const void *table_scratch, int64_t n_live,
                                  void *buckets, int32_t *payload_sorted, const float *tile_zcut, int32_t *long_flag,
                                  int rel_gate, fsgs_stream_t stream) {
    if (rel_gate < 0) return FSGS_EINVAL;
    return bin_live_emit_impl(C, N, means2d, radii, depths, conics, opacities, tile_width, tile_height, isect_offsets,
                              table_scratch, n_live, buckets, nullptr, nullptr, payload_sorted, stream, tile_zcut,
                              long_flag, rel_gate);
}

static int bin_live_emit_impl(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                              const float *conics, const float *opacities, int tile_width, int tile_height,
                              const int32_t *isect_offsets, const void *table_scratch, int64_t n_live,
                              void *buckets, void *buckets2, void *split_scratch, int32_t *payload_sorted,
                              fsgs_stream_t stream, const float *tile_zcut, int32_t *long_flag, int rel_gate) {
    if (C < 1 || N < 0 || tile_width < 1 || tile_height < 1 || n_live < 0 || !isect_offsets) return FSGS_EINVAL;
    if (tile_zcut && C != 1) return FSGS_EINVAL;
    const int64_t T64 = (int64_t)C * tile_width * tile_height;
    if (T64 > fsgs_bin_live_max_tiles() || n_live > 0x7FFFFFF0ll) return FSGS_EINVAL;
    if (n_live == 0 || (int64_t)C * N == 0) return FSGS_OK;
    if (!means2d || !radii || !depths || !conics || !opacities || !table_scratch || !buckets || !payload_sorted)
        return FSGS_EINVAL;
    const int T = (int)T64;
    const int n_tiles = tile_width * tile_height;
    hipStream_t s = as_stream(stream);
    const int nb = (int)bin_blocks((int64_t)C * N);
    int32_t *table = const_cast<int32_t *>(reinterpret_cast<const int32_t *>(table_scratch));
    BinProjArgs pj{};
    size_t extra_lds = 0;
    bin_place_zcut(pj, tile_zcut, tile_width, tile_height, T, (int64_t)C * N, extra_lds);
    int rc = bin_launch<true, false, false>(extra_lds, T, (int64_t)C * N, nb, s, C, N, means2d, radii, depths, conics,
                                            opacities, tile_width, tile_height, T, nb, bin_chunks((int64_t)C * N),
                                            (int32_t *)nullptr, table, isect_offsets,
                                            reinterpret_cast<uint64_t *>(buckets), pj, (int)n_live);
    if (rc != FSGS_OK) return rc;
    int tb = 0;
    while ((1ll << tb) <= n_tiles) ++tb;
    if (tb < 1) tb = 1;
    if (buckets2)
        return launch_split_sort(T, n_tiles, tb, isect_offsets, n_live, reinterpret_cast<const uint64_t *>(buckets),
                                 reinterpret_cast<uint64_t *>(buckets2), reinterpret_cast<int32_t *>(split_scratch),
                                 payload_sorted, s);
    return launch_tile_sort_tiers(T, n_tiles, tb, isect_offsets, reinterpret_cast<uint64_t *>(buckets),
                                  payload_sorted, nullptr, s, long_flag, rel_gate);
}


// ---- count-free binning (round 5): ONE enumeration instead of two ----------------------------------------------------
// fsgs_project_bin_live_count_sh_pack + fsgs_bin_live_emit enumerate every (Gaussian, tile) pair twice — once to count
// per tile, once to fill — with a [workgroup][tile] table, its column scan and the offsets in between.  When the caller
// knows how much room every tile's bucket needs (it does: the same VIEW's previous frame, plus a margin), the pairs can
// be dropped into buckets with room to spare in ONE pass: a workgroup projects its 1024 Gaussians, counts its live pairs
// per tile in LDS, claims that many words in every touched bucket with one returning atomic per tile, and walks the same
// pairs again (their reach masks kept in two registers) to fill.  The SH / packing launch follows with ONE extra
// workgroup that turns the claimed counts into isect_offsets (exact, compact), the NEXT frame's bucket bases and the
// verdict: n_live_mapped[3] = 1 if a tile outgrew its bucket — then the frame must be redone through the exact two-pass
// route (nothing else has consumed the lists yet: the caller looks before the backward, as it does for the capacity).
// fsgs_bin_live_sort_buckets then sorts every bucket from its place in `buckets` into the compact payload list.
//   bucket_base [T + 1] i32 (in): bucket_base[T] <= bucket_words; tile_cursor [T] i32: zero on entry, zero on return;
//   next_bucket_base [T + 1] i32 (out): exclusive sums of  count * growth + slack.
// One camera, N such that a workgroup takes one chunk (N <= 1024 x 65535), no occlusion cuts (dense scenes keep the
// two-pass route), fp32 attributes.  Lists are bit-identical to the two-pass route's (the in-tile sort orders them).
extern "C" int fsgs_project_bin_live_fill_sh_pack(
    int N, const float *means, const float *quats, const float *log_scales, float *opac_logit, int binarise,
    float binary_threshold, const float *viewmat, const float *K, int width, int height, float eps2d, float near_plane,
    float far_plane, float radius_clip, float *scales_out, float *opac_out, int32_t *radii, float *means2d,
    float *depths, float *conics, int tile_width, int tile_height, int32_t *tiles_per_gauss, int32_t *isect_offsets,
    const int32_t *bucket_base, int32_t *tile_cursor, void *buckets, int64_t bucket_words, int32_t *next_bucket_base,
    float growth, int slack, int32_t *n_live_mapped, int degree, const float *campos, const float *features_dc,
    const float *features_rest, const float *c2w, float *packed, float *normals_world, float *zero_cells, int n_zero,
    int32_t *tile_order, fsgs_stream_t stream) {
    if (N < 1 || tile_width < 1 || tile_height < 1 || !isect_offsets || !bucket_base || !tile_cursor || !buckets ||
        !next_bucket_base || bucket_words < 1 || bucket_words >= 0x40000000ll || growth < 1.f || slack < 0)
        return FSGS_EINVAL;
    const int64_t T64 = (int64_t)tile_width * tile_height;
    if (T64 > fsgs_bin_live_max_tiles()) return FSGS_EINVAL;
    const int T = (int)T64;
    if (!means || !quats || !log_scales || !opac_logit || !viewmat || !K || !scales_out || !opac_out || !radii ||
        !means2d || !depths || !conics || !campos || !features_dc || !features_rest || !c2w || !packed || !normals_world ||
        degree < 0 || degree > 3)
        return FSGS_EINVAL;
    const int64_t total = N;
    if (bin_chunks(total) != 1) return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
    // (more threads than Gaussians in a workgroup — see isect_live_bin_body)
    const int gpw = bin_threads(total);
    const int bt = gpw * (gpw == 256 ? FSGS_FILL_MULT_256 : (gpw == 512 ? FSGS_FILL_MULT_512 : 1));
    const int nb = (int)((total + gpw - 1) / gpw);
    BinProjArgs pj = {means, quats, log_scales, viewmat, K, opac_logit, scales_out, opac_out, means2d, depths,
                      conics, radii, width, height, binarise, eps2d, near_plane, far_plane, radius_clip,
                      binary_threshold, nullptr, 0, nullptr, 0, 0, 0, nullptr, nullptr, bucket_base, tile_cursor};
    const size_t need = (bt == 1024 ? sizeof(BinLds<1024>) : (bt == 512 ? sizeof(BinLds<512>) : sizeof(BinLds<256>))) +
                        (size_t)T * sizeof(int);
    uint64_t *bk = reinterpret_cast<uint64_t *>(buckets);
#define FSGS_BIN_FILL(BT)                                                                                              \
    do {                                                                                                               \
        if (const int rc = ensure_dynamic_lds<&isect_live_bin_fill_kernel<false, BT>>(need)) return rc;                \
        hipLaunchKernelGGL((isect_live_bin_fill_kernel<false, BT>), dim3(nb), dim3(BT), need, s, N, tile_width,         \
                           tile_height, T, nb, tiles_per_gauss, bk, pj, (int)bucket_words, gpw);                        \
    } while (0)
    switch (bt) {
    case 1024: FSGS_BIN_FILL(1024); break;
    case 512: FSGS_BIN_FILL(512); break;
    default: FSGS_BIN_FILL(256); break;
    }
#undef FSGS_BIN_FILL
    int rc = check_launch();
    if (rc != FSGS_OK) return rc;
    const ShPackRider r = {N, degree, 0, means, campos, features_dc, features_rest, radii, depths, means2d, conics,
                           opac_out, quats, log_scales, c2w, packed, normals_world, zero_cells, n_zero, nullptr};
    return launch_scan_rows_sh_pack(T, nb, nullptr, nullptr, r, s, nullptr, isect_offsets, n_live_mapped,
                                    BucketBook{tile_cursor, bucket_base, next_bucket_base, growth, slack, tile_order});
}

// the in-tile sorts of the count-free route: bucket t = `buckets`[bucket_base[t] ..) with isect_offsets[t + 1] -
// isect_offsets[t] words -> payload_sorted[isect_offsets[t] ..)  (long_flag / rel_gate as in fsgs_bin_live_emit)
extern "C" int fsgs_bin_live_sort_buckets(int tile_width, int tile_height, const int32_t *isect_offsets,
                                          const int32_t *bucket_base, void *buckets, int32_t *payload_sorted,
                                          int32_t *long_flag, int rel_gate, const int32_t *tile_order,
                                          fsgs_stream_t stream) {
    if (tile_width < 1 || tile_height < 1 || !isect_offsets || !bucket_base || !buckets || !payload_sorted || rel_gate < 0)
        return FSGS_EINVAL;
    const int n_tiles = tile_width * tile_height;
    if (n_tiles > fsgs_bin_live_max_tiles()) return FSGS_EINVAL;
    int tb = 0;
    while ((1ll << tb) <= n_tiles) ++tb;
    if (tb < 1) tb = 1;
    return launch_tile_sort_tiers(n_tiles, n_tiles, tb, isect_offsets, reinterpret_cast<uint64_t *>(buckets),
                                  payload_sorted, nullptr, as_stream(stream), long_flag, rel_gate, bucket_base, tile_order);
}

// ---- occlusion cut: next frame's tile cuts from this frame's walk, and the verdict on this frame's own cuts --------
// One 64-thread workgroup per tile.  tile_open[t] != 0: some pixel of the tile was still transparent when its list
// ended (set by fsgs_raster_fwd_quad) -> no cut for this tile next time (+inf), and if the frame was binned WITH a finite
// cut for it the frame is invalid (entries behind the cut could have contributed).  Otherwise every pixel stopped at or
// before the tile's deepest composited entry: the cut is that entry's depth plus a margin (half the walked depth range,
// at least 1 %): entries behind it were not reached by this frame and will not be binned by the next frame of this view.
__global__ void __launch_bounds__(64)
tile_zcut_kernel(int tw, int th, int W, int H, const int32_t *__restrict__ last_ids, const int32_t *__restrict__ payload,
                 const float *__restrict__ depths, const int32_t *__restrict__ offsets, int32_t *__restrict__ tile_open,
                 const float *__restrict__ zcut_in, int32_t *__restrict__ cand, float *__restrict__ zcut_out,
                 int32_t *__restrict__ bad, float margin_span, float margin_rel, float tail_frac) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const int ty = t / tw, tx = t - ty * tw;
    int best = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pidx = lane * 4 + k, i = ty * 16 + (pidx >> 4), j = tx * 16 + (pidx & 15);
        if (i < H && j < W) best = max(best, last_ids[(int64_t)i * W + j]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
    if (lane != 0) return;
    const int l0 = offsets[t], l1 = offsets[t + 1];
    const int open = tile_open[t];
    tile_open[t] = 0;  // (left cleared for the next frame's forward)
    const float inf = __builtin_huge_valf();
    float z = inf;
    if (open) {
        // an open tile under a cut: the frame is exact only if no LIVE pair of the tile fell behind the cut — to be
        // looked at by fsgs_tile_zcut_recheck (rare: the tile saturated in front of the cut one visit ago)
        if (zcut_in && zcut_in[t] < inf) {
            cand[t] = 1;
            atomicOr(bad, 2);
        }
    } else if (l1 > l0) {
        const int last = min(max(best, l0), l1 - 1);
        const float z_last = depths[payload[last] & 0x0FFFFFFF], z_first = depths[payload[l0] & 0x0FFFFFFF];
        z = z_last + fmaxf(margin_span * (z_last - z_first), margin_rel * z_last);
        // a tile that needed (nearly) all of its list to saturate gains nothing from a cut and is the tile that will
        // be found open one visit later: no cut for it (FSGS_ZCUT_TAIL: the fraction of the list that must lie behind)
        if ((float)(l1 - 1 - last) < tail_frac * (float)(l1 - l0)) z = inf;
    } else if (zcut_in) {
        z = zcut_in[t];  // (an empty list under a cut: keep the cut; without a cut an empty tile needs none)
    }
    zcut_out[t] = z;
}

// the verdict into host-mapped memory once every tile has been looked at: [0] <- bad, [1] <- 1 ("landed")
__global__ void zcut_verdict_kernel(int32_t *__restrict__ bad, int32_t *__restrict__ verdict_mapped,
                                    int32_t *__restrict__ clear, int n_clear) {
    for (int t = threadIdx.x; t < n_clear; t += blockDim.x) clear[t] = 0;
    if (threadIdx.x != 0) return;
    verdict_mapped[0] = *bad;
    *bad = 0;
    __threadfence_system();
    __hip_atomic_store(&verdict_mapped[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One camera.  last_ids [H,W] and payload / isect_offsets [T + 1] as the forward used them, depths [N], tile_open [T]
// (int32, written by fsgs_raster_fwd_quad, cleared here), zcut_in (nullable: the cuts this frame was binned with),
// zcut_cand ([T] int32, zeroed; required with zcut_in), zcut_out [T], bad_scratch: one zeroed int32 on the device (left
// zeroed), verdict_mapped: int32[2] in host-mapped memory ([1] cleared by the caller beforehand).
extern "C" int fsgs_tile_zcut_update(int tile_width, int tile_height, int width, int height, const int32_t *last_ids,
                                     const int32_t *payload, const float *depths, const int32_t *isect_offsets,
                                     int32_t *tile_open, const float *zcut_in, int32_t *zcut_cand, float *zcut_out,
                                     int32_t *bad_scratch, int32_t *verdict_mapped, float margin_span, float margin_rel,
                                     float tail_frac, fsgs_stream_t stream) {
    if (tile_width < 1 || tile_height < 1 || width < 1 || height < 1 || margin_span < 0.f || margin_rel < 0.f ||
        !(tail_frac >= 0.f && tail_frac <= 1.f))
        return FSGS_EINVAL;
    if (!last_ids || !payload || !depths || !isect_offsets || !tile_open || !zcut_out || !bad_scratch || !verdict_mapped ||
        (zcut_in && !zcut_cand))
        return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(tile_zcut_kernel, dim3(tile_width * tile_height), dim3(64), 0, s, tile_width, tile_height, width,
                       height, last_ids, payload, depths, isect_offsets, tile_open, zcut_in, zcut_cand, zcut_out,
                       bad_scratch, margin_span, margin_rel, tail_frac);
    hipLaunchKernelGGL(zcut_verdict_kernel, dim3(1), dim3(64), 0, s, bad_scratch, verdict_mapped, (int32_t *)nullptr, 0);
    return check_launch();
}

// The second look at a frame whose verdict was 2: does any candidate tile (open, under a finite cut) have a LIVE pair
// behind its cut?  One thread per Gaussian over its gsplat rectangle, the binning's own reach test on candidate tiles.
__global__ void __launch_bounds__(256)
zcut_recheck_kernel(int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                    const float *__restrict__ depths, const float *__restrict__ conics,
                    const float *__restrict__ opacities, int tw, int th, const float *__restrict__ zcut_in,
                    const int32_t *__restrict__ cand, int32_t *__restrict__ bad) {
    extern __shared__ unsigned char cand_l[];
    for (int t = threadIdx.x; t < tw * th; t += 256) cand_l[t] = cand[t] ? 1 : 0;
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int r = radii[n];
    if (r <= 0) return;
    const float2 m = reinterpret_cast<const float2 *>(means2d)[n];
    const TileRect t = tile_rect(m.x, m.y, r, 16, tw, th, 0);
    bool any = false;
    for (int y = t.y0; y < t.y1 && !any; ++y)
        for (int x = t.x0; x < t.x1; ++x) any = any || cand_l[y * tw + x];
    if (!any) return;
    const float z = depths[n];
    const CullPrep cp = cull_prepare(m.x, m.y, opacities[n], conics[n * 3 + 0], conics[n * 3 + 1], conics[n * 3 + 2]);
    for (int y = t.y0; y < t.y1; ++y)
        for (int x = t.x0; x < t.x1; ++x)
            if (cand_l[y * tw + x] && z > zcut_in[y * tw + x] && quadrant_mask(cp, (float)(x * 16), (float)(y * 16))) {
                *bad = 1;  // (same value from every writer)
                return;
            }
}

// verdict_mapped[0] <- 1 if a candidate tile of zcut_cand lost a live pair to zcut_in (the frame is INVALID) else 0,
// [1] <- 1 once landed ([1] cleared by the caller beforehand); zcut_cand and bad_scratch are left zeroed.  means2d /
// radii / depths / conics / opacities [N]: the frame's projection outputs (one camera).
extern "C" int fsgs_tile_zcut_recheck(int N, const float *means2d, const int32_t *radii, const float *depths,
                                      const float *conics, const float *opacities, int tile_width, int tile_height,
                                      const float *zcut_in, int32_t *zcut_cand, int32_t *bad_scratch,
                                      int32_t *verdict_mapped, fsgs_stream_t stream) {
    if (N < 0 || tile_width < 1 || tile_height < 1 || !zcut_in || !zcut_cand || !bad_scratch || !verdict_mapped)
        return FSGS_EINVAL;
    if (N > 0 && (!means2d || !radii || !depths || !conics || !opacities)) return FSGS_EINVAL;
    const int T = tile_width * tile_height;
    if (T > fsgs_bin_live_max_tiles()) return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
    if (N > 0)
        hipLaunchKernelGGL(zcut_recheck_kernel, dim3(ceil_div(N, 256)), dim3(256), (size_t)T, s, N, means2d, radii, depths,
                           conics, opacities, tile_width, tile_height, zcut_in, zcut_cand, bad_scratch);
    hipLaunchKernelGGL(zcut_verdict_kernel, dim3(1), dim3(64), 0, s, bad_scratch, verdict_mapped, zcut_cand, T);
    return check_launch();
}
