// E8 on live lists: the VJP of front-to-back alpha compositing over the per-quadrant record streams
// that fsgs_raster_fwd_quad (raster_quad.hip) leaves behind.  Same arithmetic and the same results as
// the generic tile kernel in raster.hip (replaces gsplat 1.0.0
// `rasterize_to_pixels_bwd` for /root/reference/dn_splatter/dn_model.py:570-591 and the legacy
// rasterize_backward for :644-653), for tile_size == 16.
//
// A workgroup of four waves owns one 8x8 quadrant; wave w takes the 64-record segments w, w+4, ... of
// the quadrant's stream independently (the forward saved every pixel's state at each segment end).
// Per segment each lane loads one record (three or four coalesced 16-byte loads), parks it in its
// wave's LDS slot, and the wave iterates over the records with broadcast LDS reads, one pixel per lane.
// The 64 per-pixel partials of the 15 gradient components are summed with a TRANSPOSED row reduction
// (common.h: 33 DPP adds instead of 60), the four rows meet through two lane permutes, and lanes
// 0..14 issue ONE global_atomic_add_f32 into a packed 64-byte per-Gaussian record (one cache line),
// instead of 15 single-lane atomics to six different arrays.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "cull.h"

namespace fsgs {

// E = number of "extra" channels (0 or 3) composited in the same walk from the record's 4th
// float4 — FusionSense's normal plane: background 1, and its gradient must not reach the projected
// centres (xys are detached at /root/reference/dn_splatter/dn_model.py:638) nor absgrad.
template <int E>
struct QLds {
    float4 r0[64], r1[64], r2[64], r3[E ? 64 : 1];
    uint8_t list[4][64];  // per 4x4 pixel block: the segment's slots that can reach it, back to front
};

struct Rec {
    float4 r0, r1, r2, r3;
};

// A lane past the end of the segment reads the segment's first record instead (always in range) and
// zeroes it: an all-zero record composites to alpha = 0.  (No branch: a struct filled on two paths is
// kept in scratch memory by the compiler.)
template <int E>
__device__ __forceinline__ void load_rec(Rec &r, const float4 *__restrict__ rec, int64_t p, int64_t p_safe, bool ok) {
    const float4 *src = rec + (E ? 4 : 3) * (ok ? p : p_safe);
    auto keep = [ok](float4 v) {
        return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    };
    r.r0 = keep(src[0]); r.r1 = keep(src[1]); r.r2 = keep(src[2]);
    if (E) r.r3 = keep(src[3]);
}

// packed gradient record, 16 floats per (camera, Gaussian):
//   [0..3] v_colors  [4..6] v_conics  [7..8] v_means2d  [9..10] v_means2d_abs  [11] v_opacities
//   [12..14] v_extra (normal plane)
//
// Segment-parallel backward.  The forward saved every pixel's (T, accumulated colour) before each
// 64-entry segment of its quadrant list, so the two recurrences of the back-to-front walk can be
// restarted at any segment end:  T_end(k) = T_before(k+1)  (or the final T),  and the colour still
// to come  buffer_end(k) = C_total - C_before(k+1)  (or 0).  A workgroup of kBwdWaves waves owns a
// quadrant; wave w takes segments w, w+kBwdWaves, ... independently (no barriers between waves), which
// removes the long-list tail that otherwise sets the kernel's duration.
#ifndef FSGS_BWD_MERGE_THR16
#define FSGS_BWD_MERGE_THR16 15
#endif
#ifndef FSGS_BWD_WAVES
#define FSGS_BWD_WAVES 4
#endif
constexpr int kBwdWaves = FSGS_BWD_WAVES;
// waves per SIMD the register allocator aims at (78 VGPRs -> 6).  Round 6, measured (make EXTRA=-DFSGS_BWD_OCC=n, config #2,
// it/s of two runs each): 5 -> 2043 / 2059, 6 -> 2040 / 2042, 7 (5 spills) -> 2033 / 2018, 8 (13 spills) -> 1994 / 1989 — and
// a software-pipelined walk (the next step's list entry and record loaded from LDS one step ahead, +12 v_mov per step, 89-95
// VGPRs at 5 waves): 2053 / 2043 and 2041 / 2043 against 2049 / 2045: the walk is bound by vector ISSUE, not
// by the latency of its LDS reads.
#ifndef FSGS_BWD_OCC
#define FSGS_BWD_OCC 6
#endif
// waves per SIMD the register allocator aims at (78 VGPRs -> 6; A/B: make EXTRA=-DFSGS_BWD_OCC=n)
#ifndef FSGS_BWD_OCC
#define FSGS_BWD_OCC 6
#endif

// Optional: the gradients arrive as those of FusionSense's get_outputs images (D = 4, E = 3, C = 1; what
// fsgs_epilogue_bwd would first turn into v_render / v_alphas / v_render_extra, dn_model.py:602-613, 655-664):
//   rgb = clamp(render.rgb + (1 - alpha) bg, 0, 1),  depth = alpha > 0 ? render.w : fill,
//   normal = (normalize(render_extra) + 1) / 2.
// Every lane derives its pixel's gradients itself (pointwise, ~40 instructions per wave): one launch and a
// write + read of 32 B per pixel less.
struct GetOutputsGrads {
    const float *v_rgb, *v_depth, *v_normal, *v_alpha_in, *bg;  // v_rgb == nullptr: not used
};

#ifdef FSGS_BWD_STATS
// Diagnostic build only (make EXTRA=-DFSGS_BWD_STATS OUT=../libfsgs_stats.so; tools/bwd_lane_stats.py): how full the
// 64 lanes of a step are.  [0] loop iterations, [1] iterations executed (some lane valid), [2] lanes with a list entry
// in executed iterations, [3] lanes that pass the alpha test in executed iterations, [4] segments, [5] segments walked
// merged, [6] sum of the segments' longest row list, [7] sum of the segments' union sizes, and — what a finer pixel group
// would need — [8] sum over segments of max over the 8 half-rows (4x2 pixels) of the steps in which that half-row had a
// valid lane, [9] the same for the 16 quads (2x2 pixels), [10] / [11] (step, half-row) / (step, quad) pairs with a valid lane.
__device__ unsigned long long g_bwd_stats[12];
#endif

template <int D, bool ABS, int E>
__global__ void __launch_bounds__(64 * kBwdWaves) __attribute__((amdgpu_waves_per_eu(FSGS_BWD_OCC, FSGS_BWD_OCC)))
raster_bwd_live_kernel(int64_t cap, const float4 *__restrict__ rec,
                       const int32_t *__restrict__ tile_offsets, int64_t n_isects,
                       const float *__restrict__ backgrounds, int W, int H, int tw, int th,
                       const float *__restrict__ render, const float *__restrict__ alphas,
                       const int32_t *__restrict__ last_ids, const float *__restrict__ v_render,
                       const float *__restrict__ v_alphas, const float *__restrict__ seg_state,
                       int64_t seg_cap, float *__restrict__ v_packed, int normalize_last,
                       const float *__restrict__ render_extra, const float *__restrict__ v_render_extra,
                       const int32_t *__restrict__ n_rec, GetOutputsGrads ep, int64_t replica_rows, int merge_thr16,
                       int perm_stride, const int32_t *__restrict__ seg_split, const int32_t *__restrict__ bwd_queue,
                       int bwd_qcap, const int32_t *__restrict__ bwd_order) {
    __shared__ QLds<E> Lw[kBwdWaves];
    constexpr int RS = E ? 4 : 3;
    constexpr int SS = 64 * (1 + D + E);
    const int cam = blockIdx.z;
    const int n_tiles_total = gridDim.z * th * tw;
    // (perm_stride > 1: workgroups that are dispatched together take quadrants that lie `stride` apart instead of
    // neighbours, so that their atomics land on different Gaussians' gradient lines: fsgs_set_bwd_dispatch_stride)
    int bx = blockIdx.x, by = blockIdx.y;
    // Rows beyond the image's quadrant rows are EXTRA workgroups (single camera only): each takes one queued group of
    // segments of a long quadrant, whose own workgroup keeps the segments from seg_split on (raster_quad.hip:
    // split_backward) — a list of 44 segments is walked by six workgroups side by side instead of 11 rounds of one.
    int seg_lo = 0, seg_hi = -1;  // (-1: the quadrant's own workgroup: [seg_split, n_seg))
    // (the extra rows come FIRST in the dispatch order: they hold the segments of the launch's longest streams)
    const int extra_rows = (int)gridDim.y - 2 * th;
    const int grid_row = (int)blockIdx.y - extra_rows;
    if (grid_row < 0) {
        const int item = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
        if (!bwd_queue || item >= min(bwd_queue[0], bwd_qcap)) return;
        const int4 it = reinterpret_cast<const int4 *>(bwd_queue + 4)[item];
        const int t = it.x & 0x0FFFFFFF, qq = (it.x >> 28) & 3;
        const int ty = t / tw, tx = t - ty * tw;
        bx = 2 * tx + (qq & 1); by = 2 * ty + (qq >> 1);
        seg_lo = it.y; seg_hi = it.z;
    } else if (bwd_order) {
        // longest own share first (raster_quad.hip: split_backward files every quadrant under a class of record counts):
        // workgroup b of the image rows takes entry b of the classes laid end to end, the longest class first
        constexpr int kOrderClasses = 32;
        const int b = grid_row * (int)gridDim.x + (int)blockIdx.x;
        const int lane_o = threadIdx.x & 63;
        const int32_t *cnt = bwd_queue + 4 + 4 * bwd_qcap;
        const int mine = lane_o < kOrderClasses ? cnt[kOrderClasses - 1 - lane_o] : 0;  // lane l: class 31 - l
        int incl = mine;
#pragma unroll
        for (int d = 1; d < kOrderClasses; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane_o >= d) incl += o;
        }
        const uint64_t past = __ballot(lane_o < kOrderClasses && incl > b);  // classes whose end lies beyond b
        if (past == 0ull) return;  // (b >= the quadrants filed: nothing left — quadrants the forward never finished)
        const int l = __ffsll((long long)past) - 1;
        const int excl = __builtin_amdgcn_readlane(incl - mine, l);
        const unsigned lin = (unsigned)bwd_order[(int64_t)(kOrderClasses - 1 - l) * (gridDim.x * 2 * th) + (b - excl)];
        by = lin / gridDim.x; bx = lin - by * gridDim.x;
    } else if (perm_stride > 1) {
        const unsigned total = gridDim.x * (unsigned)(2 * th);
        const unsigned lin = (unsigned)(((unsigned long long)((unsigned)grid_row * gridDim.x + blockIdx.x) * (unsigned)perm_stride) % total);
        by = lin / gridDim.x; bx = lin - by * gridDim.x;
    } else {
        by = grid_row;
    }
    const int tile_x = bx >> 1, tile_y = by >> 1;
    const int q = ((by & 1) << 1) | (bx & 1);
    const int tile_lin = (cam * th + tile_y) * tw + tile_x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    QLds<E> &L = Lw[w];
    // row r of the wave (16 lanes) owns the 4x4 pixel block (r >> 1, r & 1) of the quadrant and walks ITS OWN
    // list of the segment's records (those that can reach the block): a record reaches 2.6 of a quadrant's 4
    // blocks on average, so a 64-record segment costs max_r(n_r) ~ 45 steps instead of 64
    const int row = lane >> 4, pl = lane & 15;
    const int p = (((row >> 1) << 2) + (pl >> 2)) * 8 + ((row & 1) << 2) + (pl & 3);  // row-major in the quadrant
    const int j = bx * 8 + (p & 7), i = by * 8 + (p >> 3);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const bool inside = (i < H) && (j < W);
    const int64_t pix_id = ((int64_t)cam * H + min(i, H - 1)) * W + min(j, W - 1);

    const int64_t l0 = tile_offsets[tile_lin];
    // streams written by fsgs_raster_fwd_quad: stream position l0 + 4 * tile, record count from the forward
    const int s = (int)l0 + 4 * tile_lin;
    const int e = s + n_rec[(int64_t)q * n_tiles_total + tile_lin];
    const int n_seg = (e - s + 63) >> 6;
    if (seg_hi < 0) {
        seg_hi = n_seg;
        seg_lo = seg_split ? seg_split[(int64_t)q * n_tiles_total + tile_lin] : 0;
    }
    if (w >= seg_hi - seg_lo) return;
    const float4 *stream = rec + RS * ((int64_t)q * cap);
    const float *seg_q = seg_state + (int64_t)q * seg_cap * SS;

    const float alpha_px = alphas[pix_id];
    const float T_final = 1.f - alpha_px;
    float v_out[D], c_total[D];
#pragma unroll
    for (int k = 0; k < D; ++k) c_total[k] = render[pix_id * D + k];
    float v_out_a = 0.f;
    const bool from_images = (D == 4 && E == 3) && ep.v_rgb != nullptr;
    if (from_images) {
        float va = (inside && ep.v_alpha_in) ? ep.v_alpha_in[pix_id] : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float pre = c_total[k] + T_final * ep.bg[k];
            const float gk = (inside && pre >= 0.f && pre <= 1.f) ? ep.v_rgb[pix_id * 3 + k] : 0.f;
            v_out[k] = gk;
            va -= ep.bg[k] * gk;
        }
        v_out[D - 1] = (inside && ep.v_depth && alpha_px > 0.f) ? ep.v_depth[pix_id] : 0.f;
        v_out_a = va;
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) v_out[k] = inside ? v_render[pix_id * D + k] : 0.f;
        v_out_a = inside ? v_alphas[pix_id] : 0.f;
    }
    if (normalize_last) {
        // render[D-1] = acc / max(alpha, 1e-10): undo for the colour total, chain the two gradients
        const float a_c = fmaxf(alpha_px, 1e-10f);
        const float ed = c_total[D - 1];
        c_total[D - 1] = ed * a_c;
        const float v_ed = v_out[D - 1];
        v_out[D - 1] = v_ed / a_c;
        if (alpha_px > 1e-10f) v_out_a -= v_ed * ed / a_c;
    }
    if (backgrounds) {
#pragma unroll
        for (int k = 0; k < D; ++k) c_total[k] -= T_final * backgrounds[cam * D + k];
    }
    float v_oute[E ? E : 1], ce_total[E ? E : 1];
    float bge_dot = 0.f;
    if (from_images) {
        const float nx = render_extra[pix_id * 3 + 0], ny = render_extra[pix_id * 3 + 1], nz = render_extra[pix_id * 3 + 2];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (inside && ep.v_normal) {
            const float inv = 1.f / sqrtf(nx * nx + ny * ny + nz * nz);
            const float ux = nx * inv, uy = ny * inv, uz = nz * inv;
            const float hx = 0.5f * ep.v_normal[pix_id * 3 + 0], hy = 0.5f * ep.v_normal[pix_id * 3 + 1],
                        hz = 0.5f * ep.v_normal[pix_id * 3 + 2];
            const float d = hx * ux + hy * uy + hz * uz;
            gx = (hx - d * ux) * inv; gy = (hy - d * uy) * inv; gz = (hz - d * uz) * inv;
        }
        const float gg[3] = {gx, gy, gz};
        const float nn[3] = {nx, ny, nz};
#pragma unroll
        for (int k = 0; k < E; ++k) {
            v_oute[k] = gg[k];
            ce_total[k] = nn[k] - T_final;  // background = 1
            bge_dot += v_oute[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) {
            v_oute[k] = inside ? v_render_extra[pix_id * E + k] : 0.f;
            ce_total[k] = render_extra[pix_id * E + k] - T_final;  // background = 1
            bge_dot += v_oute[k];
        }
    }
    float bg_dot = 0.f;
    if (backgrounds) {
#pragma unroll
        for (int k = 0; k < D; ++k) bg_dot += backgrounds[cam * D + k] * v_out[k];
    }
    const int bin_final = inside ? last_ids[pix_id] : -1;
    int rbf = bin_final;  // last composited list index over the row's 16 pixels
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) rbf = max(rbf, __shfl_xor(rbf, d, 64));
    int row_bin_final[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) row_bin_final[r4] = __builtin_amdgcn_readlane(rbf, 16 * r4);
    const float quad_x0 = (float)(bx * 8), quad_y0 = (float)(by * 8);
    // this workgroup's replica of a large Gaussian's gradient line (common.h: grad_spread); 0 rows = no replicas
    // (neighbouring tiles and the four 8x8 quadrants of a tile — one workgroup each — take different replicas)
    // (32-bit float offsets into v_packed: the launcher guarantees the accumulator stays below 2^32 bytes)
    const uint32_t rep_off = (uint32_t)replica_rows * 16u * ((((unsigned)tile_lin * 2654435761u >> 16) + (bx & 1) +
                                                             2 * (by & 1)) % kGradReplicas);

#ifdef FSGS_BWD_STATS
    unsigned long long st_iter = 0, st_exec = 0, st_have = 0, st_valid = 0, st_seg = 0, st_merge = 0, st_long = 0, st_union = 0;
    unsigned long long st_half_max = 0, st_quad_max = 0, st_half_pairs = 0, st_quad_pairs = 0;
#endif
    for (int seg = seg_hi - 1 - w; seg >= seg_lo; seg -= kBwdWaves) {
        const int b0 = s + (seg << 6);          // first stream position of the segment
        const int n = min(64, e - b0);
        // slot t holds stream position b0 + n - 1 - t (descending list order)
        Rec r;
        load_rec<E>(r, stream, (int64_t)b0 + n - 1 - lane, (int64_t)b0, lane < n);
        L.r0[lane] = r.r0; L.r1[lane] = r.r1; L.r2[lane] = r.r2;
        if (E) L.r3[lane] = r.r3;
        // which blocks this lane's record can reach (exact in the conservative direction, cull.h; padding
        // records have opacity 0 and reach nothing), and whether any of the block's pixels composited it
        unsigned bm = 0u;
        if (lane < n) {
            const CullPrep cp = cull_prepare(r.r0.x, r.r0.y, r.r0.z, r.r0.w, r.r1.x, r.r1.y);
            bm = block_mask(cp, quad_x0, quad_y0);
        }
        const int isect_l = __float_as_int(r.r1.z);
        int my_n = 0, steps = 0;
        uint64_t mrow[4];
        unsigned rm = 0u;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const bool lv = ((bm >> r4) & 1u) && (isect_l <= row_bin_final[r4]);
            mrow[r4] = __ballot(lv);
            rm |= lv ? (1u << r4) : 0u;
            const int c = __popcll(mrow[r4]);
            my_n = (row == r4) ? c : my_n;
            steps = max(steps, c);
        }
        // A segment whose longest row list is (nearly) as long as the UNION of the four (records of large footprint
        // reach every block) gains nothing from separate lists: all four rows then walk the union in lockstep, the
        // rows' totals of a record meet through two lane-permute adds, and ONE 64-byte atomic per (quadrant, record)
        // leaves instead of up to four — on the hot gradient lines of large Gaussians, which the memory side
        // serialises, a quarter of the transactions.  Wave-uniform choice per segment.
        const uint64_t many = mrow[0] | mrow[1] | mrow[2] | mrow[3];
        const int n_union = __popcll(many);
        const bool merge = n_union > 0 && steps * 16 >= n_union * merge_thr16;
#ifdef FSGS_BWD_STATS
        st_seg += 1; st_merge += merge ? 1 : 0; st_long += steps; st_union += n_union;
        int half_cnt = 0, quad_cnt = 0;  // steps of this segment in which this lane's half-row / quad had a valid lane
#endif
        const uint64_t below = (1ull << lane) - 1ull;
        if (merge) {
            if (rm) L.list[0][__popcll(many & below)] = (uint8_t)lane;
            my_n = steps = n_union;
        } else {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                if ((rm >> r4) & 1u) L.list[r4][__popcll(mrow[r4] & below)] = (uint8_t)lane;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Per pixel the walk needs the colour still behind the current record only through its product with
        // the pixel's output gradient:  sum_k (c_k T - buffer_k ra) v_k = T (c . v) - ra (buffer . v),  and
        // buffer . v grows by fac (c . v) per record.  One running scalar per plane instead of D (+E) channel
        // accumulators: 4 instead of 5 instructions per channel, 5 fewer live registers.
        float T = T_final, Bd = 0.f, Be = 0.f;
        if (steps > 0 && seg != n_seg - 1) {
            const float *slot = seg_q + ((int64_t)((b0 + 64) >> 6) + tile_lin) * SS;
            T = slot[p];
#pragma unroll
            for (int k = 0; k < D; ++k) Bd += (c_total[k] - slot[64 * (1 + k) + p]) * v_out[k];
#pragma unroll
            for (int k = 0; k < E; ++k) Be += (ce_total[k] - slot[64 * (1 + D + k) + p]) * v_oute[k];
        }
        auto walk = [&](auto merge_tag) {
        constexpr bool MERGE = decltype(merge_tag)::value;
        for (int kk = 0; kk < steps; ++kk) {
            const bool have = kk < my_n;
            const int t = have ? (int)L.list[MERGE ? 0 : row][kk] : 0;
            const float4 a0 = L.r0[t], a1 = L.r1[t];
            const int isect = __float_as_int(a1.z);
            bool valid = have && inside && (isect <= bin_final);
            const float dx = a0.x - px, dy = a0.y - py;
            const float sigma = 0.5f * (a0.w * dx * dx + a1.y * dy * dy) + a1.x * dx * dy;
            const float vis0 = __expf(-sigma);
            if (sigma < 0.f || fminf(kAlphaMax, a0.z * vis0) < kAlphaMin) valid = false;
#ifdef FSGS_BWD_STATS
            st_iter += 1;
            if (__any(valid)) {
                st_exec += 1; st_have += __popcll(__ballot(have)); st_valid += __popcll(__ballot(valid));
                const uint64_t vb = __ballot(valid);
                // lane = 16 row + pl, pl = 4 y + x inside the 4x4 block: half-row = pl >> 3 (two pixel rows), quad = 2x2
                const int hshift = (lane & ~7);
                half_cnt += ((vb >> hshift) & 0xFFull) ? 1 : 0;
                const int qy = (pl >> 3), qx = (pl >> 1) & 1;  // quad (qy, qx) holds pl in {8 qy + 2 qx + {0, 1, 4, 5}}
                const uint64_t qmask = (0x33ull << (2 * qx + 8 * qy)) << (16 * row);
                quad_cnt += (vb & qmask) ? 1 : 0;
                for (int hh = 0; hh < 8; ++hh) st_half_pairs += ((vb >> (8 * hh)) & 0xFFull) ? 1 : 0;
                for (int r4 = 0; r4 < 4; ++r4)
                    for (int qq = 0; qq < 4; ++qq)
                        st_quad_pairs += (vb & ((0x33ull << (2 * (qq & 1) + 8 * (qq >> 1))) << (16 * r4))) ? 1 : 0;
            }
#endif
            if (!__any(valid)) continue;

            // Branch-free from here: a lane that does not contribute gets vis = 0, hence alpha = 0,
            // ra = 1 (T unchanged), fac = 0 and v_sigma = 0 — every term below vanishes by itself.
            const float vis = valid ? vis0 : 0.f;
            const float alpha = fminf(kAlphaMax, a0.z * vis);
            float g[12], ge[E ? E : 1];
            const float4 c4 = L.r2[t];
            const float col[4] = {c4.x, c4.y, c4.z, c4.w};
            const float ra = __builtin_amdgcn_rcpf(1.f - alpha);  // 1 ulp; alpha <= 0.999
            T *= ra;
            const float fac = alpha * T;
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = 0.f;
            float Cd = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                g[k] = fac * v_out[k];
                Cd += col[k] * v_out[k];
            }
            float v_alpha = T * Cd - ra * Bd;
            Bd += fac * Cd;
            v_alpha += T_final * ra * v_out_a;
            if (backgrounds) v_alpha -= T_final * ra * bg_dot;
            float v_alpha_e = 0.f;  // the extra plane's share: conics / opacity only
            if (E) {
                const float4 e4 = L.r3[t];
                const float ce[3] = {e4.x, e4.y, e4.z};
                float Ce = 0.f;
#pragma unroll
                for (int k = 0; k < E; ++k) {
                    ge[k] = fac * v_oute[k];
                    Ce += ce[k] * v_oute[k];
                }
                v_alpha_e = T * Ce - ra * Be - T_final * ra * bge_dot;
                Be += fac * Ce;
            }
            {
                const float ov = (a0.z * vis <= kAlphaMax) ? a0.z * vis : 0.f;  // clamp active: no sigma grads
                const float v_sigma_xy = -ov * v_alpha;
                const float v_sigma = -ov * (v_alpha + v_alpha_e);
                g[4] = 0.5f * v_sigma * dx * dx;
                g[5] = v_sigma * dx * dy;
                g[6] = 0.5f * v_sigma * dy * dy;
                g[7] = v_sigma_xy * (a0.w * dx + a1.x * dy);
                g[8] = v_sigma_xy * (a1.x * dx + a1.y * dy);
                g[9] = ABS ? fabsf(g[7]) : 0.f;
                g[10] = ABS ? fabsf(g[8]) : 0.f;
                g[11] = (a0.z * vis <= kAlphaMax) ? vis * (v_alpha + v_alpha_e) : 0.f;
            }
            // 15 per-pixel partials -> the row's 15 per-Gaussian totals: transposed row sum (lane l ends with
            // the sum of value l & 15 over its row's 16 pixels); lanes 0..14 of every row add their record's
            // totals with ONE atomic instruction (four Gaussians, one cache line each)
            float vals[16];
#pragma unroll
            for (int k = 0; k < 12; ++k) vals[k] = g[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) vals[12 + k] = (k < E) ? ge[k] : 0.f;
            vals[15] = 0.f;
            // not merged: lane l ends with its row's total of value l & 15;  merged: lane (row, column c) with the
            // WAVE's total of value 4 row + (c >> 2) (common.h)
            float tot;
            int vi;
            bool writer;
            if (MERGE) {
                tot = wave_transpose_sum16(vals);
                vi = 4 * row + (pl >> 2);
                writer = (pl & 3) == 0;
            } else {
                row_transpose_sum16(vals);
                tot = vals[0];
                vi = pl;
                writer = true;
            }
            if (writer && vi < 12 + E && tot != 0.f) {
                const bool used = (vi < D) || (vi >= 4 && vi <= 8) || (ABS && (vi == 9 || vi == 10)) ||
                                  (vi == 11) || (vi >= 12);
                if (used) {
                    const uint32_t gid = __float_as_uint(a1.w) & 0x0FFFFFFFu;
                    const uint32_t off = grad_spread(a0.w, a1.x, a1.y) ? rep_off : 0u;
                    unsafeAtomicAdd(v_packed + (off + gid * 16u + (uint32_t)vi), tot);
                }
            }
        }
        };
        if (merge) walk(std::true_type{}); else walk(std::false_type{});
#ifdef FSGS_BWD_STATS
        {
            int hm = half_cnt, qm = quad_cnt;
            for (int d = 32; d >= 1; d >>= 1) { hm = max(hm, __shfl_xor(hm, d, 64)); qm = max(qm, __shfl_xor(qm, d, 64)); }
            st_half_max += hm; st_quad_max += qm;
        }
#endif
        __builtin_amdgcn_wave_barrier();  // LDS reads of this segment precede the next segment's writes
    }
#ifdef FSGS_BWD_STATS
    if (lane == 0) {
        const unsigned long long v[12] = {st_iter, st_exec, st_have, st_valid, st_seg, st_merge, st_long, st_union,
                                          st_half_max, st_quad_max, st_half_pairs, st_quad_pairs};
        for (int k = 0; k < 12; ++k) atomicAdd(&g_bwd_stats[k], v[k]);
    }
#endif
}

__global__ void __launch_bounds__(256)
unpack_grads_kernel(int64_t total, int D, float4 *__restrict__ v_packed, int rezero, float *__restrict__ v_means2d,
                    float *__restrict__ v_means2d_abs, float *__restrict__ v_conics,
                    float *__restrict__ v_colors, float *__restrict__ v_opacities,
                    float *__restrict__ v_extra, float *__restrict__ v_last) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= total) return;
    const float4 a = v_packed[n * 4 + 0], b = v_packed[n * 4 + 1], c = v_packed[n * 4 + 2];
    if (v_last) v_last[n] = (D == 4) ? a.w : ((D == 3) ? a.z : a.x);  // gradient of the last colour channel
    const float4 d = v_packed[n * 4 + 3];
    if (rezero) {  // leave the accumulator ready for the next frame's atomics (saves a 64 B/Gaussian fill launch)
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        v_packed[n * 4 + 0] = zero4; v_packed[n * 4 + 1] = zero4; v_packed[n * 4 + 2] = zero4; v_packed[n * 4 + 3] = zero4;
    }
    if (v_extra) {
        v_extra[n * 3 + 0] = d.x; v_extra[n * 3 + 1] = d.y; v_extra[n * 3 + 2] = d.z;
    }
    if (D == 4) {
        reinterpret_cast<float4 *>(v_colors)[n] = a;
    } else {
        v_colors[n * D + 0] = a.x;
        if (D > 1) v_colors[n * D + 1] = a.y;
        if (D > 2) v_colors[n * D + 2] = a.z;
    }
    v_conics[n * 3 + 0] = b.x; v_conics[n * 3 + 1] = b.y; v_conics[n * 3 + 2] = b.z;
    reinterpret_cast<float2 *>(v_means2d)[n] = make_float2(b.w, c.x);
    if (v_means2d_abs) reinterpret_cast<float2 *>(v_means2d_abs)[n] = make_float2(c.y, c.z);
    v_opacities[n] = c.w;
}

}  // namespace fsgs

using namespace fsgs;

extern "C" int64_t fsgs_quad_stream_capacity(int C, int tile_width, int tile_height, int64_t n_isects);
extern "C" int64_t fsgs_quad_seg_slots(int C, int tile_width, int tile_height, int64_t n_isects);

// Dispatch order of the quadrants (0 / 1: row-major; k > 1: consecutive workgroups take quadrants k apart, made coprime
// with the number of quadrants).  Neighbouring quadrants share their Gaussians: dispatched together they add to the same
// gradient lines at the same time.  Measured (ms of the launch, row-major vs stride 7): config #2 0.181 / 0.168, config #4
// 0.720 / 0.611, config #5 0.703 / 0.610 — and config #3 0.209 / 0.279 (there the lines of the large hull Gaussians stay
// L2-resident under the row-major sweep).  Not decided here: the caller measures (fused.py tunes it per frame shape).
// The stride is an ARGUMENT of the two launch entry points (dispatch_stride >= 0).  DEPRECATED: a negative argument takes
// this process-wide default (fsgs_set_bwd_dispatch_stride; initially row-major), kept for callers written against round 3.
static std::atomic<int> g_bwd_dispatch_stride{0};
extern "C" int fsgs_set_bwd_dispatch_stride(int stride) { return g_bwd_dispatch_stride.exchange(stride < 0 ? 0 : stride); }

#ifdef FSGS_BWD_STATS
extern "C" int fsgs_debug_bwd_stats(unsigned long long *out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(fsgs::g_bwd_stats), sizeof(unsigned long long) * 12) != hipSuccess) return -2;
    if (reset) {
        const unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(fsgs::g_bwd_stats), z, sizeof(z)) != hipSuccess) return -2;
    }
    return 0;
}
#endif

static int launch_bwd_live(int C, int D, const float *records, const int32_t *n_rec,
                           int64_t cap, int64_t seg_cap, const int32_t *isect_offsets, int64_t n_isects,
                           const float *backgrounds, int width, int height, int tile_width, int tile_height,
                           int normalize_last, const float *render, const float *alphas, const int32_t *last_ids,
                           const float *v_render, const float *v_alphas, const float *seg_state, int with_abs,
                           const float *render_extra, const float *v_render_extra, float *v_packed,
                           fsgs_stream_t stream, int dispatch_stride,
                           GetOutputsGrads ep = GetOutputsGrads{nullptr, nullptr, nullptr, nullptr, nullptr},
                           int64_t replica_rows = 0, const int32_t *seg_split = nullptr,
                           const int32_t *bwd_queue = nullptr, int bwd_queue_items = 0,
                           const int32_t *bwd_order = nullptr) {
    // FSGS_BWD_MERGE_THR16 (build macro, make EXTRA=-DFSGS_BWD_MERGE_THR16=n): a segment walks the union list with merged
    // atomics when its longest row list is >= thr/16 of the union; 0 = always, 17 = never
    constexpr int merge_thr16 = FSGS_BWD_MERGE_THR16;
    if (C < 0 || width < 0 || height < 0 || n_isects < 0) return FSGS_EINVAL;
    if (C == 0 || width == 0 || height == 0 || n_isects == 0) return FSGS_OK;
    // (the kernel addresses the accumulator with 32-bit float offsets: record ids are 28-bit, 16 floats each, plus the
    // replica lines — (2^28 + kGradReplicas * replica_rows) * 16 floats must stay below 2^32 bytes / 4)
    if (replica_rows > (int64_t)1 << 24) return FSGS_EINVAL;
    if (!records || !n_rec || !isect_offsets || !render || !alphas || !last_ids || !seg_state || !v_packed)
        return FSGS_EINVAL;
    if (ep.v_rgb) {
        if (C != 1 || D != 4 || !render_extra || !ep.bg) return FSGS_EINVAL;
    } else if (!v_render || !v_alphas) {
        return FSGS_EINVAL;
    }
    hipStream_t s = as_stream(stream);
    if (bwd_queue_items < 0 || (bwd_queue && (!seg_split || C != 1))) return FSGS_EINVAL;
    if (bwd_order && !bwd_queue) return FSGS_EINVAL;
    // (extra rows of workgroups, one per queue item: see the kernel)
    const int extra_rows = (bwd_queue && bwd_queue_items > 0) ? ceil_div(bwd_queue_items, 2 * tile_width) : 0;
    const dim3 grid(2 * tile_width, 2 * tile_height + extra_rows, C);
    const float4 *rec = reinterpret_cast<const float4 *>(records);
    int perm_stride = dispatch_stride >= 0 ? dispatch_stride : g_bwd_dispatch_stride.load(std::memory_order_relaxed);
    if (perm_stride > 1) {  // coprime with the number of quadrants
        const long long total = 4ll * tile_width * tile_height;
        auto gcd = [](long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a; };
        while (gcd(perm_stride, total) != 1) ++perm_stride;
    }
#define FSGS_BWD_LIVE(DD, AA, EE)                                                                                \
    hipLaunchKernelGGL((raster_bwd_live_kernel<DD, AA, EE>), grid, dim3(64 * kBwdWaves), 0, s, cap, rec,          \
                       isect_offsets, n_isects, backgrounds, width, height, tile_width, tile_height, render,      \
                       alphas, last_ids, v_render, v_alphas, seg_state, seg_cap, v_packed, normalize_last,         \
                       render_extra, v_render_extra, n_rec, ep, replica_rows, merge_thr16, perm_stride, seg_split,  \
                       bwd_queue, bwd_queue_items, bwd_order)
    if (render_extra) {
        if (D != 4 || (!v_render_extra && !ep.v_rgb)) return FSGS_EINVAL;
        if (with_abs) FSGS_BWD_LIVE(4, true, 3); else FSGS_BWD_LIVE(4, false, 3);
        return check_launch();
    }
    switch (D) {
        case 1: if (with_abs) FSGS_BWD_LIVE(1, true, 0); else FSGS_BWD_LIVE(1, false, 0); break;
        case 3: if (with_abs) FSGS_BWD_LIVE(3, true, 0); else FSGS_BWD_LIVE(3, false, 0); break;
        case 4: if (with_abs) FSGS_BWD_LIVE(4, true, 0); else FSGS_BWD_LIVE(4, false, 0); break;
        default: return FSGS_EINVAL;
    }
#undef FSGS_BWD_LIVE
    return check_launch();
}

extern "C" int fsgs_raster_bwd_quad(int C, int D, const float *records, const int32_t *n_rec,
                                    const int32_t *isect_offsets, int64_t n_isects, const float *backgrounds,
                                    int width, int height, int tile_width, int tile_height, int normalize_last,
                                    const float *render, const float *alphas, const int32_t *last_ids,
                                    const float *v_render, const float *v_alphas, const float *seg_state,
                                    int with_abs, const float *render_extra, const float *v_render_extra,
                                    float *v_packed, int dispatch_stride, const int32_t *seg_split,
                                    const int32_t *bwd_queue, int bwd_queue_items, const int32_t *bwd_order,
                                    fsgs_stream_t stream) {
    if (n_isects > 0 && !n_rec) return FSGS_EINVAL;
    return launch_bwd_live(C, D, records, n_rec,
                           fsgs_quad_stream_capacity(C, tile_width, tile_height, n_isects),
                           fsgs_quad_seg_slots(C, tile_width, tile_height, n_isects), isect_offsets, n_isects,
                           backgrounds, width, height, tile_width, tile_height, normalize_last, render, alphas,
                           last_ids, v_render, v_alphas, seg_state, with_abs, render_extra, v_render_extra,
                           v_packed, stream, dispatch_stride, GetOutputsGrads{nullptr, nullptr, nullptr, nullptr, nullptr},
                           0, seg_split, bwd_queue, bwd_queue_items, bwd_order);
}

extern "C" int fsgs_raster_unpack_grads(int64_t total, int D, float *v_packed, int rezero, float *v_means2d,
                                        float *v_means2d_abs, float *v_conics, float *v_colors,
                                        float *v_opacities, float *v_extra, float *v_last,
                                        fsgs_stream_t stream) {
    if (total < 0 || (D != 1 && D != 3 && D != 4)) return FSGS_EINVAL;
    if (total == 0) return FSGS_OK;
    if (!v_packed || !v_means2d || !v_conics || !v_colors || !v_opacities) return FSGS_EINVAL;
    hipLaunchKernelGGL(unpack_grads_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), total, D,
                       reinterpret_cast<float4 *>(v_packed), rezero, v_means2d, v_means2d_abs, v_conics, v_colors,
                       v_opacities, v_extra, v_last);
    return check_launch();
}

// fsgs_epilogue_bwd + fsgs_raster_bwd_quad (one camera, RGB + expected depth and the normal plane in one walk): the
// gradients of the get_outputs images go in directly; v_depth / v_normal / v_alpha_in may be NULL (no gradient).
extern "C" int fsgs_raster_bwd_quad_images(const float *records, const int32_t *n_rec, const int32_t *isect_offsets,
                                           int64_t n_isects, int width, int height, int tile_width, int tile_height,
                                           const float *render, const float *alphas, const int32_t *last_ids,
                                           const float *render_extra, const float *background, const float *v_rgb,
                                           const float *v_depth, const float *v_normal, const float *v_alpha_in,
                                           const float *seg_state, int with_abs, float *v_packed,
                                           int64_t replica_rows, int dispatch_stride, const int32_t *seg_split,
                                           const int32_t *bwd_queue, int bwd_queue_items, const int32_t *bwd_order,
                                           fsgs_stream_t stream) {
    if (n_isects > 0 && (!n_rec || !v_rgb || !background)) return FSGS_EINVAL;
    if (!v_rgb || replica_rows < 0) return FSGS_EINVAL;
    const GetOutputsGrads ep = {v_rgb, v_depth, v_normal, v_alpha_in, background};
    return launch_bwd_live(1, 4, records, n_rec, fsgs_quad_stream_capacity(1, tile_width, tile_height, n_isects),
                           fsgs_quad_seg_slots(1, tile_width, tile_height, n_isects), isect_offsets, n_isects, nullptr,
                           width, height, tile_width, tile_height, 1, render, alphas, last_ids, nullptr, nullptr,
                           seg_state, with_abs, render_extra, nullptr, v_packed, stream, dispatch_stride, ep, replica_rows,
                           seg_split, bwd_queue, bwd_queue_items, bwd_order);
}
