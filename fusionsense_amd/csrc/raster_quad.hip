// E7 forward on live lists, second generation: the culling filter, the per-quadrant compaction and the
// compositing in ONE kernel (replaces gsplat 1.0.0 `rasterize_to_pixels_fwd` for
// /root/reference/dn_splatter/dn_model.py:570-591 and the legacy rasterize_forward for :644-653).
//
// What the first generation (live.hip scan4 + compact, raster_live.hip forward) measured on BASELINE
// config #2, per-wave timestamps over one launch:
//   * 2.43 M quadrant records were written by the compaction pass, 1.17 M were ever composited (the
//     rest sit behind the point where every pixel of the quadrant is opaque);
//   * one wave walked a quadrant's list alone at ~400 cycles per record (LDS read -> sigma -> exp ->
//     T is a dependent chain), so the kernel lasted as long as its longest list (7 x 64 records,
//     ~90 us) while the machine stood 2/3 empty.
// Here a workgroup of four waves owns one 8x8 quadrant:
//   * it reads the tile's sorted list itself (payload = quadrant mask << 28 | Gaussian id, written
//     by fsgs_isect_emit_live), keeps the entries whose mask has this quadrant's bit, gathers their
//     64-byte line from the packed per-Gaussian table and parks them in LDS in list order — only as
//     far as the pixels are still transparent; the records are streamed out once for the backward;
//   * a wave composites 16 pixels x 4 consecutive records per step: lane = (pixel, record slot),
//     the transmittance chain across the four slots is a quad prefix product (two DPP multiplies),
//     so a list of n records costs n/4 dependent steps instead of n.
// Arithmetic per record is that of the reference (alpha = min(.999, o*exp(-sigma)), skip < 1/255,
// stop before T <= 1e-4); products/sums over the four slots associate differently from a strictly
// sequential walk (<= 1 ulp per step), which the float tolerance of the parity tests covers.
#include "common.h"

namespace fsgs {

constexpr int kMaxCells = 256;               // partial image maxima (fsgs_raster_quad_max_cells)
constexpr int kQuadChunk = 256;             // list entries examined per step by the workgroup
constexpr int kQuadStage = kQuadChunk + 4;  // staged records (+ padding to a multiple of 4)

// Long lists in the BACKWARD (round 5): its workgroup of four waves takes a quadrant's 64-record segments four at a time,
// so a quadrant of 44 segments (config #3's hull tiles) is 11 dependent rounds while the launch's other workgroups have
// long left (0.21 ms at 0.62 of the vector-ALU cycles).  The wave that finishes a quadrant's forward therefore splits a
// stream of more than kBwdOwnSegs segments: the quadrant's own workgroup keeps the LAST kBwdOwnSegs (seg_split = where
// they start), the segments in front of them are queued in groups of kBwdOwnSegs for extra workgroups of the backward
// launch (fsgs_raster_bwd_quad*: bwd_queue).  The queue's counter is zeroed by the caller (it rides behind the max_last
// cells); a queue without room leaves the quadrant whole (seg_split = 0).
constexpr int kBwdOwnSegs = 8;
__device__ __forceinline__ void split_backward(int tile_lin, int q, int n_tiles_total, int n_records,
                                               int32_t *__restrict__ seg_split, int32_t *__restrict__ bwd_queue,
                                               int bwd_qcap) {  // (one lane)
    if (!seg_split) return;
    const int n_seg = (n_records + 63) >> 6;
    int split = 0;
    // (a queue that is already full is not asked again: in a dense frame every quadrant qualifies, and tens of thousands
    // of returning atomics on one word would queue up behind each other)
    if (bwd_queue && n_seg > kBwdOwnSegs &&
        __hip_atomic_load(bwd_queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bwd_qcap) {
        const int rest = n_seg - kBwdOwnSegs, groups = (rest + kBwdOwnSegs - 1) / kBwdOwnSegs;
        const int slot = atomicAdd(bwd_queue, groups);
        if (slot + groups <= bwd_qcap) {
            int4 *items = reinterpret_cast<int4 *>(bwd_queue + 4);
            for (int g = 0; g < groups; ++g) {
                const int hi = rest - kBwdOwnSegs * g;
                items[slot + g] = make_int4(tile_lin | (q << 28), max(hi - kBwdOwnSegs, 0), hi, 0);
            }
            split = rest;
        }
    }
    seg_split[(int64_t)q * n_tiles_total + tile_lin] = split;
}

template <int E>
struct QuadLds {
    float4 r0[kQuadStage], r1[kQuadStage], r2[kQuadStage], r3[E ? kQuadStage : 1];
    int wcount[4];
    int wdone[4];
};

// quad_perm DPP controls: lane i of every 4-lane group reads lane sel[i]
#define FSGS_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int CTRL>
__device__ __forceinline__ float quad_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int quad_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ float quad_sum(float v) {
    v += quad_f<FSGS_QP(1, 0, 3, 2)>(v);
    v += quad_f<FSGS_QP(2, 3, 0, 1)>(v);
    return v;
}

template <int D, int E>
__global__ void __launch_bounds__(256)
raster_fwd_quad_kernel(int64_t cap, const float4 *__restrict__ packed, const int32_t *__restrict__ payload,
                       const int32_t *__restrict__ tile_offsets, int64_t n_isects,
                       const float *__restrict__ backgrounds, int W, int H, int tw, int th, int n_tiles_total,
                       float *__restrict__ render, float *__restrict__ alphas, int32_t *__restrict__ last_ids,
                       float4 *__restrict__ rec_out, int32_t *__restrict__ n_rec,
                       float *__restrict__ seg_state, int64_t seg_cap, int normalize_last,
                       float *__restrict__ render_extra, float *__restrict__ max_last, int ends_on_device,
                       int32_t *__restrict__ tile_open, int32_t *__restrict__ seg_split,
                       int32_t *__restrict__ bwd_queue, int bwd_qcap) {
    __shared__ QuadLds<E> S;
    constexpr int RS = E ? 4 : 3;
    // workgroup b runs on XCD b % 8: the four quadrants of a tile share its list and its Gaussians,
    // so they are given ids that differ by 8 (same XCD, dispatched together)
    const unsigned bid = blockIdx.x;
    const int tile_lin = (bid >> 5) * 8 + (bid & 7);
    const int q = (bid >> 3) & 3;
    if (tile_lin >= n_tiles_total) return;
    const int cam = tile_lin / (tw * th);
    const int tile_in = tile_lin - cam * tw * th;
    const int tile_y = tile_in / tw, tile_x = tile_in - tile_y * tw;
    const int qx = 2 * tile_x + (q & 1), qy = 2 * tile_y + (q >> 1);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, slot = lane & 3;
    // wave w composites the 4x4 pixel block (w >> 1, w & 1) of the quadrant (a compact footprint: fewer
    // records reach any of its pixels than would reach an 8x2 strip, so more steps skip their second half);
    // p = the pixel's row-major index in the 8x8 quadrant, the order the backward expects
    const int pl = lane >> 2;
    const int p = (((w >> 1) << 2) + (pl >> 2)) * 8 + ((w & 1) << 2) + (pl & 3);
    const int j = qx * 8 + (p & 7), i = qy * 8 + (p >> 3);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const bool inside = (i < H) && (j < W);
    bool done = !inside;

    const int l0 = tile_offsets[tile_lin];
    // (ends_on_device: tile_offsets holds n_tiles_total + 1 entries, the last one = the live total as the binning
    // left it on the device — the caller did not wait for it and passed a CAPACITY as n_isects)
    const int l1 = (tile_lin == n_tiles_total - 1 && !ends_on_device) ? (int)n_isects : tile_offsets[tile_lin + 1];
    // A quadrant's records go to stream q from position sbase on.  Padding can exceed the tile's own
    // entry count by at most 3 (last chunk only), hence the 4 spare positions per tile.
    const int64_t sbase = (int64_t)l0 + 4ll * tile_lin;
    float4 *stream = rec_out ? rec_out + RS * ((int64_t)q * cap + sbase) : nullptr;
    constexpr int SS = 64 * (1 + D + E);  // floats per segment-state slot
    float *seg_q = seg_state ? seg_state + (int64_t)q * seg_cap * SS : nullptr;

    float T = 1.f;
    int32_t cur_idx = 0;
    float pix[D], pxe[E ? E : 1];  // this lane's share (its record slot) of the pixel's sums
#pragma unroll
    for (int k = 0; k < D; ++k) pix[k] = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) pxe[k] = 0.f;

    // one list entry per thread, fetched one chunk ahead of the compositing
    bool live = false;
    float4 f0, f1, f2, f3;
    auto fetch = [&](int c0) {
        const int idx = c0 + tid;
        live = false;
        if (idx < l1) {
            const uint32_t pay = (uint32_t)payload[idx];
            if ((pay >> (28 + q)) & 1u) {
                live = true;
                const int32_t g = (int32_t)(pay & 0x0FFFFFFFu);
                const float4 *src = packed + (int64_t)g * 4;
                f0 = src[0]; f1 = src[1]; f2 = src[2];
                if (E) f3 = src[3];
                f1.z = __int_as_float(idx);
                f1.w = __int_as_float(g);
            }
        }
    };
    fetch(l0);

    int cnt = 0;  // records composited so far (always a multiple of 4; padding records included)
    for (int c0 = l0; c0 < l1; c0 += kQuadChunk) {
        const uint64_t bal = __ballot(live);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        const int wave_done = __all(done) ? 1 : 0;  // (a vote must not sit inside the lane-0 branch)
        if (lane == 0) {
            S.wcount[w] = __popcll(bal);
            S.wdone[w] = wave_done;
        }
        __syncthreads();  // counts visible; the previous chunk's LDS reads are finished
        if (S.wdone[0] & S.wdone[1] & S.wdone[2] & S.wdone[3]) break;  // every pixel is opaque
        int base = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = S.wcount[k];
            base += (k < w) ? c : 0;
            total += c;
        }
        const int n_proc = (total + 3) & ~3;
        if (live) {
            const int pos = base + before;
            S.r0[pos] = f0; S.r1[pos] = f1; S.r2[pos] = f2;
            if (E) S.r3[pos] = f3;
            if (stream) {
                float4 *dst = stream + RS * (int64_t)(cnt + pos);
                dst[0] = f0; dst[1] = f1; dst[2] = f2;
                if (E) dst[3] = f3;
            }
        }
        // padding to a whole group: zero opacity (alpha = 0) and the list index of the chunk's last
        // record, so the stream stays sorted by list index for the backward's range tests
        if (tid >= total && tid < n_proc) {
            const int last_idx = min(c0 + kQuadChunk, l1) - 1;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 z1 = make_float4(0.f, 0.f, __int_as_float(last_idx), __int_as_float(0));
            S.r0[tid] = z; S.r1[tid] = z1; S.r2[tid] = z;
            if (E) S.r3[tid] = z;
            if (stream) {
                float4 *dst = stream + RS * (int64_t)(cnt + tid);
                dst[0] = z; dst[1] = z1; dst[2] = z;
                if (E) dst[3] = z;
            }
        }
        if (c0 + kQuadChunk < l1) fetch(c0 + kQuadChunk);  // in flight while this chunk is composited
        __syncthreads();  // staged records visible

        if (!__all(done)) {
            for (int t = 0; t < n_proc; t += 4) {
                const float4 a0 = S.r0[t + slot], a1 = S.r1[t + slot];
                const float dx = a0.x - px, dy = a0.y - py;
                const float sigma = 0.5f * (a0.w * dx * dx + a1.y * dy * dy) + a1.x * dx * dy;
                const float alpha = fminf(kAlphaMax, a0.z * __expf(-sigma));
                const bool pass = !done && !(sigma < 0.f) && !(alpha < kAlphaMin);
                if (__any(pass)) {
                    // transmittance before / after this lane's record: quad prefix product of (1 - alpha)
                    const float om = pass ? 1.f - alpha : 1.f;
                    float inc = om;
                    const float s1 = quad_f<FSGS_QP(0, 0, 1, 2)>(inc);
                    inc = (slot >= 1) ? inc * s1 : inc;
                    const float s2 = quad_f<FSGS_QP(0, 1, 0, 1)>(inc);
                    inc = (slot >= 2) ? inc * s2 : inc;
                    const float e1 = quad_f<FSGS_QP(0, 0, 1, 2)>(inc);
                    const float exc = (slot >= 1) ? e1 : 1.f;
                    // a record that would take T to <= 1e-4 ends the pixel and is not composited; every
                    // later record then fails the same test (T only falls), so the test is per lane
                    const bool stop = pass && (T * inc <= kTMin);
                    const bool commit = pass && !stop;
                    const float vis = commit ? alpha * (T * exc) : 0.f;
                    const float4 c = S.r2[t + slot];
                    pix[0] += c.x * vis;
                    if (D > 1) pix[1] += c.y * vis;
                    if (D > 2) pix[2] += c.z * vis;
                    if (D > 3) pix[D - 1] += c.w * vis;
                    if (E) {
                        const float4 ce = S.r3[t + slot];
                        pxe[0] += ce.x * vis;
                        if (E > 1) pxe[1] += ce.y * vis;
                        if (E > 2) pxe[E - 1] += ce.z * vis;
                    }
                    cur_idx = commit ? __float_as_int(a1.z) : cur_idx;
                    float f = commit ? om : 1.f;
                    f *= quad_f<FSGS_QP(1, 0, 3, 2)>(f);
                    f *= quad_f<FSGS_QP(2, 3, 0, 1)>(f);
                    T *= f;
                    int st = stop ? 1 : 0;
                    st |= quad_i<FSGS_QP(1, 0, 3, 2)>(st);
                    st |= quad_i<FSGS_QP(2, 3, 0, 1)>(st);
                    done = done || (st != 0);
                }
                // state of every pixel after each 64-record segment, for the segment-parallel backward
                if (seg_q && ((cnt + t + 4) & 63) == 0) {
                    float *sl = seg_q + (((sbase + cnt + t + 4) >> 6) + tile_lin) * SS;
                    float tot[D + (E ? E : 0)];
#pragma unroll
                    for (int k = 0; k < D; ++k) tot[k] = quad_sum(pix[k]);
#pragma unroll
                    for (int k = 0; k < E; ++k) tot[D + k] = quad_sum(pxe[k]);
                    if (slot == 0) {
                        sl[p] = T;
#pragma unroll
                        for (int k = 0; k < D + E; ++k) sl[64 * (1 + k) + p] = tot[k];
                    }
                }
            }
        } else if (seg_q) {
            // a finished wave still owes the segment states of the records the others walk
            for (int t = 0; t < n_proc; t += 4) {
                if (((cnt + t + 4) & 63) == 0) {
                    float *sl = seg_q + (((sbase + cnt + t + 4) >> 6) + tile_lin) * SS;
                    float tot[D + (E ? E : 0)];
#pragma unroll
                    for (int k = 0; k < D; ++k) tot[k] = quad_sum(pix[k]);
#pragma unroll
                    for (int k = 0; k < E; ++k) tot[D + k] = quad_sum(pxe[k]);
                    if (slot == 0) {
                        sl[p] = T;
#pragma unroll
                        for (int k = 0; k < D + E; ++k) sl[64 * (1 + k) + p] = tot[k];
                    }
                }
            }
        }
        cnt += n_proc;
    }
    if (n_rec && tid == 0) {
        n_rec[(int64_t)q * n_tiles_total + tile_lin] = cnt;
        split_backward(tile_lin, q, n_tiles_total, cnt, seg_split, bwd_queue, bwd_qcap);
    }
    // (occlusion cut, isect.hip: fsgs_tile_zcut_update) a pixel that is still transparent at the end of the list makes
    // its tile "open": this frame needed — or would have needed — everything the tile had
    if (tile_open && !__all(done) && lane == 0) tile_open[tile_lin] = 1;

    // a pixel's sums are spread over its four slot lanes
#pragma unroll
    for (int k = 0; k < D; ++k) pix[k] = quad_sum(pix[k]);
#pragma unroll
    for (int k = 0; k < E; ++k) pxe[k] = quad_sum(pxe[k]);
    cur_idx = max(cur_idx, quad_i<FSGS_QP(1, 0, 3, 2)>(cur_idx));
    cur_idx = max(cur_idx, quad_i<FSGS_QP(2, 3, 0, 1)>(cur_idx));

    if (backgrounds) {
#pragma unroll
        for (int k = 0; k < D; ++k) pix[k] += T * backgrounds[cam * D + k];
    }
    if (normalize_last) pix[D - 1] = pix[D - 1] / fmaxf(1.f - T, 1e-10f);  // expected depth
    if (inside && slot == 0) {
        const int64_t pix_id = ((int64_t)cam * H + i) * W + j;
        if (D == 4) {
            reinterpret_cast<float4 *>(render)[pix_id] = make_float4(pix[0], pix[1], pix[2], pix[D - 1]);
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) render[pix_id * D + k] = pix[k];
        }
        if (E) {
#pragma unroll
            for (int k = 0; k < E; ++k) render_extra[pix_id * E + k] = pxe[k] + T;  // background = 1
        }
        alphas[pix_id] = 1.f - T;
        last_ids[pix_id] = cur_idx;
    }
    if (max_last) {  // image maximum of the (normalised) last channel, for depth_im's fill value
        float m = inside ? pix[D - 1] : 0.f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        if (lane == 0) S.wcount[w] = __float_as_int(m);
        __syncthreads();
        // one atomic per workgroup, spread over kMaxCells addresses (same-address atomics serialise at
        // the memory side); non-negative floats order like their bit patterns
        if (tid == 0) {
            const int mm = max(max(S.wcount[0], S.wcount[1]), max(S.wcount[2], S.wcount[3]));
            if (mm > 0) atomicMax(reinterpret_cast<int *>(max_last) + (bid & (kMaxCells - 1)), mm);
        }
    }
}

#ifndef FSGS_FWD_WAVE_OCC
#define FSGS_FWD_WAVE_OCC 6  // waves per SIMD: 4 / 6 / 8 measured at 92.5 / 85.0 / 87.8 us (config #2), 0.598 / 0.550 / 0.562 ms (#4);
                             // with the final hot loop 5 / 6 / 7: 77.0 / 75.8 / 77.9 us, 0.527 / 0.521 / 0.534 ms; "long" from 1.5 / 2 / 3
                             // times the mean list length: 75.3 / 75.8 / 76.2 us, 0.528 / 0.521 / 0.530 ms
#endif
// Third generation (round 4), the training path's forward (D = 4, E = 3): ONE WAVE per 8x8 quadrant, lane = pixel, one
// record per step.  The (pixel, slot) walk above spends ~50 vector instructions per step of 4 records x 16 pixels — 50 per
// (quadrant, record) — of which the quad prefix products, the slot selects and the four-way reductions are pure
// bookkeeping; with one record per step and all 64 lanes on one pixel each, a (quadrant, record) costs 27 (the plain
// arithmetic of the reference's loop body).  Counters at config #2: 24.6 M vector instructions per launch against 49.1 M.
//   * the wave takes the tile's list 64 entries at a time (payloads two chunks ahead, the 64-byte lines of the entries
//     with its quadrant bit one chunk ahead), streams them out for the backward (no padding records) and parks them in
//     LDS, structure of arrays; the walk reads ONE record per step with broadcast reads (52 B per lane), the next
//     record's reads issued before this one's use, two register sets in turn;
//   * finished pixels have gate = 0 (their alpha is 0, nothing passes), the number of open pixels is a wave-uniform
//     counter, wave masks are taken from the compares themselves; the hot loop runs to the next 64-record boundary, the
//     segment-state stores sit outside it;
//   * long lists first: the grid holds every quadrant twice, the first copy walks the quadrants of tiles with more than
//     twice the mean list length, the second copy the others (the other copy leaves at once, ~3 us per launch) — a lone
//     wave advances at ~0.17 us per record whatever shares its SIMD, so the longest walks of a frame (450 records at
//     config #2: 75 us) must start at once.
// Measured (MI355X, us per launch, config #2 / #3 / #4): second generation 95 / 99 / 675, this kernel 85 / 90 / 550.
// Measured and dropped on the way (profiles/README.md): the record through the scalar cache
// (s_load_dwordx16 of its line, operands in SGPRs, 135 / 121 / 920: one line in flight per wave, and a deeper queue does
// not fit the 102 SGPRs); four records per step branch-free (115 / 138 / 980); s_setprio for the long walks (no effect);
// long-lists-first for the second-generation kernel (98: it is not bound by its tail); a hybrid launch of 256-thread
// workgroups — four waves on one quadrant of a long tile (the second generation's walk: 12.5 instructions per record on a
// wave's critical path instead of 27), one wave per quadrant of the others — 102 us at config #2 with tiles above twice
// the mean list length taken as long, 92 above four times, 84 with none (config #4: 0.62 / 0.57 / 0.54 ms): every tile
// handed to the four-wave walk costs more than its shorter chain saves; four INDEPENDENT waves per quadrant of a long
// tile (one per 4x4 block, the four-wave walk's lanes without its workgroup: each stages the list for itself, none knows
// when the quadrant has finished, so the lists are streamed to their end) — 172 / 126 / 97 us with tiles above 2 / 3 / 4
// times the mean taken as long (config #4: 1.21 / 0.97 / 0.81 ms).  What would help is in DESIGN.md 10.2.
#ifdef FSGS_FWD_TRACE
// Diagnostic build only (tools/fwd_trace.py): start / end clock, records walked, list length, chunks of every wave.
__device__ long long g_fwd_trace[1 << 17][6];
#endif
template <int E>
struct WaveLds { float4 r0[66], r1[66], r2[66], r3[E ? 66 : 1]; };  // (+2: the walk reads one record ahead)

// ---- Fourth generation (round 5): the long walks are HANDED OFF and finished chunk-parallel -------------------------
// The one-wave walk is the cheapest per record (27 vector instructions) but a quadrant's walk is one wave's dependent
// chain, so a launch ended with its longest lists (config #2: the bulk done after 45 us, the 450-record walks at 75 us;
// config #3 after densification: lists of 2 800 entries, 278 us — the four-wave walk had to be used there).  A list is a
// serial dependence only through the transmittance T, so:
//   base waves (one per quadrant, as before): a wave that has streamed `handoff_records` records and still has open
//     pixels and list entries left stops at the end of its 64-entry chunk, claims one queue position per remaining chunk
//     with ONE returning atomic, leaves its per-pixel state (T, the 7 sums, last index, open flag) and exits;
//   tail waves (the LAST `qmax` workgroups of the SAME launch, one per queue position — they are dispatched when the
//     base workgroups have all been dispatched, i.e. into the SIMD slots the launch's drain phase leaves idle, and wait
//     for their position to be filled or for every base workgroup to have left): (1) a chunk forms the product of
//     (1 - alpha) over its records for each pixel — no dependence on anything — and publishes it; (2) the transmittance at
//     its start is the handed-off T times its predecessors' products, read straight from their positions (consecutive
//     ones, held by workgroups dispatched before it); (3) it composites its records from that T with the reference's
//     stop rule (a pixel whose T at the chunk's start is <= 1e-4 stopped before it), streams them for the backward and
//     keeps chunk-local sums; (4) the wave of the list's last chunk adds the partial sums IN CHUNK ORDER
//     (deterministic), completes the saved segment states and writes the pixels.
// Same arithmetic per record; products and sums are associated per chunk (like the four-wave walk's quad products),
// covered by the image tolerance; results depend neither on the queue positions the atomics hand out nor on whether the
// queue had room (a wave whose claim does not fit walks the rest of its list itself with the chunks' arithmetic).
// Cross-workgroup traffic (cdna_hip_programming.md, Guideline 16): everything one workgroup leaves for another is stored
// write-through (sc1) and read L1-bypassing; look-back words are self-validating 8-byte granules (epoch << 32 | value),
// payloads are followed by `s_waitcnt vmcnt(0)` and a granule.  No agent-scope release fence (it writes back every dirty
// line of the XCD's L2: one per chunk took the launch from 77 to 141 us at config #2; as a second LAUNCH the tail cost
// its whole chain on top of the base waves': 78 -> 94 us at config #2, 272 -> 124 at config #3).  Every wait is
// bounded: a protocol failure sets scratch.error instead of hanging the GPU.
constexpr int kTailAreaBytes = 6144;    // per work item
constexpr int kTailHeaderBytes = 256 + 8192 + 4096;  // counters of two consecutive frames, error word, shards
constexpr int kTailMaxChunks = 1024;    // per quadrant (65 536 list entries); longer remainders are walked serially
constexpr int kTailSpinLimit = 1 << 18;
struct TailHeader {
    int32_t counter[2];   // queue positions claimed (two consecutive launches alternate)
    int32_t error;
    int32_t demand;       // (byte offset 12) lists handed off or wanting to be, ever: callers size the tail from its growth
    int32_t pad[60];
    // walking base waves that have left: 64 shards (blockIdx & 63), one 64-byte line each — one address saturates at ~88
    // atomics/us, and a line that is polled while it takes atomics stalls both (measured: 2048 waiting tail waves reading
    // these lines took the launch from 118 to 258 us).  Only the WATCHER (the last tail wave) reads them; it then sets
    // all_left, 64 granules on lines of their own, which the other tail waves poll (position & 63).
    int32_t done[2][64][16];
    unsigned long long all_left[64][8];
};
static_assert(sizeof(TailHeader) == 256 + 8192 + 4096, "tail header");
struct TailArea {
    unsigned long long cntw, donew;     // tagged: live records of the chunk; 1 + "composited something"
    unsigned long long readyw, pad1;    // tagged by the base wave: 1 = this queue position holds an item, 2 = voided
    unsigned long long P[64];           // tagged: per pixel product of (1 - alpha) over the chunk
    float base[10][64];                 // (first item of a quadrant only) pass A's state: T, 7 sums, last index, open
    int32_t base_cnt, base_pad[3];      // ... and the records it streamed
    float part[10][64];                 // chunk-local: 7 sums, last index, T after the chunk (frozen at a stop), open
};
static_assert(sizeof(TailArea) <= kTailAreaBytes, "work item area");
struct TailQueue {
    uint8_t *mem;   // [header][int4 items x qmax][areas x qmax], zeroed once by the caller; nullptr = no hand-off
    int qmax;
    uint32_t epoch; // unique per launch pair on this memory (the caller counts up from 1)
    __device__ TailHeader *header() const { return reinterpret_cast<TailHeader *>(mem); }
    __device__ int4 *items() const { return reinterpret_cast<int4 *>(mem + kTailHeaderBytes); }
    __device__ TailArea *area(int i) const {
        return reinterpret_cast<TailArea *>(mem + kTailHeaderBytes + 16ll * qmax + (int64_t)kTailAreaBytes * i);
    }
};
__device__ __forceinline__ void store_tagged(unsigned long long *p, uint32_t epoch, uint32_t v) {
    __hip_atomic_store(p, ((unsigned long long)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t wait_tagged(const unsigned long long *p, uint32_t epoch, int32_t *error) {
    for (int spin = 0; spin < kTailSpinLimit; ++spin) {
        const unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(w >> 32) == epoch) return (uint32_t)w;
        __builtin_amdgcn_s_sleep(4);
    }
    *error = 1;  // (never seen; a dispatch order that breaks the look-back's assumption would end here, not in a hang)
    return 0u;
}
// (sc1 = write-through / L1-bypassing accesses of 4-byte words: what one workgroup leaves for ANOTHER inside a launch
// goes through these and a tagged word stored after `s_waitcnt vmcnt(0)` — no agent-scope release fence, which would
// write back every dirty line of the XCD's L2 once per chunk: measured, 77 -> 141 us at config #2)
__device__ __forceinline__ void st_sc1(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The chunk arithmetic of the second pass.  ONE definition, compiled WITHOUT floating-point contraction, so that its
// results are the same bits wherever it is inlined: in the second pass's two phases (the running product of phase 3 must
// reproduce the factor phase 1 published), and in pass A's serial continuation when the queue is full — whether a list
// went through the queue or not must not show in the image (the order of the queue's atomics is not reproducible).
// alpha of one record at this lane's pixel = the reference's loop body
__device__ __forceinline__ bool tail_alpha(float px, float py, const float4 &r0, const float2 &r1, float &alpha) {
#pragma clang fp contract(off)
    const float dx = r0.x - px, dy = r0.y - py;
    const float sigma = 0.5f * (r0.w * dx * dx + r1.y * dy * dy) + r1.x * dx * dy;
    alpha = fminf(kAlphaMax, r0.z * __expf(-sigma));
    return !(sigma < 0.f) && !(alpha < kAlphaMin);
}
// phase 1: the product of (1 - alpha) over the chunk's n parked records (no stop rule)
template <int E>
__device__ __forceinline__ float tail_factor(const WaveLds<E> &S, int n, float px, float py) {
#pragma clang fp contract(off)
    float p = 1.f;
    for (int r = 0; r < n; ++r) {
        const float4 a0 = S.r0[r];
        const float2 a1 = *reinterpret_cast<const float2 *>(&S.r1[r]);
        float alpha;
        const bool ok = tail_alpha(px, py, a0, a1, alpha);
        p = ok ? p * (1.f - alpha) : p;
    }
    return p;
}
// phase 3: composite the chunk's n parked records from the transmittance T_in at its start with the reference's stop
// rule; chunk-local sums; `boundary(T_cur, pixl)` is called after record r_b (the 64-record boundary of the stream)
struct TailChunk {
    float pixl[7];     // chunk-local sums: 4 colour channels, 3 normal-plane channels
    int32_t cur_idx;   // last composited list index (0: none)
    float T_cur;       // transmittance after the chunk (frozen where the pixel stopped)
    float p_end;       // == tail_factor(...) of the same records
};
template <int E, class Boundary>
__device__ __forceinline__ void tail_composite(const WaveLds<E> &S, int n, float px, float py, float T_in, bool &open,
                                               int r_b, TailChunk &o, Boundary boundary) {
#pragma clang fp contract(off)
#pragma unroll
    for (int k = 0; k < 7; ++k) o.pixl[k] = 0.f;
    o.cur_idx = 0;
    o.T_cur = T_in;
    float p_run = 1.f;
    for (int r = 0; r < n; ++r) {
        const float4 a0 = S.r0[r];
        const float2 a1 = *reinterpret_cast<const float2 *>(&S.r1[r]);
        float alpha;
        const bool ok = tail_alpha(px, py, a0, a1, alpha);
        const float p_next = ok ? p_run * (1.f - alpha) : p_run;
        const float T_next = T_in * p_next;
        const bool low = ok && open && (T_next <= kTMin);
        const bool commit = ok && open && !low;
        const float vis = commit ? alpha * (T_in * p_run) : 0.f;
        const float4 c4 = S.r2[r], e4 = S.r3[r];
        o.pixl[0] += c4.x * vis; o.pixl[1] += c4.y * vis; o.pixl[2] += c4.z * vis; o.pixl[3] += c4.w * vis;
        o.pixl[4] += e4.x * vis; o.pixl[5] += e4.y * vis; o.pixl[6] += e4.z * vis;
        o.cur_idx = commit ? __float_as_int(e4.w) : o.cur_idx;
        o.T_cur = commit ? T_next : o.T_cur;
        open = open && !low;
        p_run = p_next;
        if (r == r_b) boundary(o.T_cur, o.pixl);
    }
    o.p_end = p_run;
}

template <int D, int E>
__device__ __forceinline__ void
raster_fwd_tail_wave(WaveLds<E> &S, const int item, const int n_base_blocks, int64_t cap,
                     const float4 *__restrict__ packed, const int32_t *__restrict__ payload,
                     const int32_t *__restrict__ tile_offsets, int64_t n_isects,
                     const float *__restrict__ backgrounds, int W, int H, int tw, int th, int n_tiles_total,
                     float *__restrict__ render, float *__restrict__ alphas, int32_t *__restrict__ last_ids,
                     float4 *__restrict__ rec_out, int32_t *__restrict__ n_rec,
                     float *__restrict__ seg_state, int64_t seg_cap, int normalize_last,
                     float *__restrict__ render_extra, float *__restrict__ max_last, int ends_on_device,
                     int32_t *__restrict__ tile_open, const TailQueue &tq, int32_t *__restrict__ seg_split,
                     int32_t *__restrict__ bwd_queue, int bwd_qcap);

template <int D, int E>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FSGS_FWD_WAVE_OCC, FSGS_FWD_WAVE_OCC)))
raster_fwd_wave_kernel(int64_t cap, const float4 *__restrict__ packed, const int32_t *__restrict__ payload,
                       const int32_t *__restrict__ tile_offsets, int64_t n_isects,
                       const float *__restrict__ backgrounds, int W, int H, int tw, int th, int n_tiles_total,
                       float *__restrict__ render, float *__restrict__ alphas, int32_t *__restrict__ last_ids,
                       float4 *__restrict__ rec_out, int32_t *__restrict__ n_rec,
                       float *__restrict__ seg_state, int64_t seg_cap, int normalize_last,
                       float *__restrict__ render_extra, float *__restrict__ max_last, int ends_on_device,
                       int32_t *__restrict__ tile_open, int handoff_records, TailQueue tq, int n_base_blocks,
                       int handoff_rel_len, int32_t *__restrict__ seg_split, int32_t *__restrict__ bwd_queue,
                       int bwd_qcap, const int32_t *__restrict__ handoff_gate, const int32_t *__restrict__ tile_order) {
    __shared__ WaveLds<E> S;
    constexpr int RS = E ? 4 : 3;
#ifdef FSGS_FWD_TRACE
    const long long t_start = wall_clock64();
#endif
    if ((int)blockIdx.x >= n_base_blocks) {
        // the launch's last workgroups: the watcher (item -1), then one per queue position (TailQueue)
        if constexpr (D == 4 && E == 3)
            raster_fwd_tail_wave<D, E>(S, (int)blockIdx.x - n_base_blocks - 1, n_base_blocks, cap, packed, payload,
                                       tile_offsets, n_isects, backgrounds, W, H, tw, th, n_tiles_total, render, alphas,
                                       last_ids, rec_out, n_rec, seg_state, seg_cap, normalize_last, render_extra,
                                       max_last, ends_on_device, tile_open, tq, seg_split, bwd_queue, bwd_qcap);
        return;
    }
    // a WALKING base wave (one of a quadrant's two copies) counts itself out when it leaves: the tail waves stop waiting
    // once all 4 x tiles have
    auto leave = [&]() {
        if (tq.mem && tq.qmax > 0 && threadIdx.x == 0)
            __hip_atomic_fetch_add(&tq.header()->done[tq.epoch & 1u][blockIdx.x & 63u][0], 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    };
    // long lists first (see above) — or, given the tiles longest first (tile_order: the count-free binning route's
    // bookkeeping leaves it), every quadrant ONCE in that order
    const unsigned half_grid = tile_order ? (unsigned)n_base_blocks : (unsigned)n_base_blocks >> 1;
    const bool long_pass = blockIdx.x < half_grid;
    const unsigned bid = long_pass ? blockIdx.x : blockIdx.x - half_grid;
    const int tile_slot = (bid >> 5) * 8 + (bid & 7);
    const int q = (bid >> 3) & 3;
    if (tile_slot >= n_tiles_total) return;
    const int tile_lin = tile_order ? tile_order[tile_slot] : tile_slot;
    {
        const int total = ends_on_device ? tile_offsets[n_tiles_total] : (int)n_isects;
        const int a0 = tile_offsets[tile_lin];
        const int a1 = (tile_lin == n_tiles_total - 1 && !ends_on_device) ? (int)n_isects : tile_offsets[tile_lin + 1];
        const bool is_long = (int64_t)(a1 - a0) * n_tiles_total > 2ll * total;
        if (!tile_order && is_long != long_pass) return;
        // only lists far longer than the frame's mean are worth queueing: they are the ones that outlast the bulk of the
        // launch (config #3's hull tiles: 2 800 entries against a mean of ~200; config #2's longest list is 5.7 x its mean
        // and saturates early — queueing those cost 10 us per launch)
        if ((int64_t)(a1 - a0) * n_tiles_total <= (int64_t)handoff_rel_len * total) handoff_records = 0;
        // ... and only in frames that have lists FAR beyond the bulk (fsgs_bin_live_emit: long_flag)
        if (handoff_gate && handoff_records > 0 && *handoff_gate == 0) handoff_records = 0;
    }
    const int cam = tile_lin / (tw * th);
    const int tile_in = tile_lin - cam * tw * th;
    const int tile_y = tile_in / tw, tile_x = tile_in - tile_y * tw;
    const int qx = 2 * tile_x + (q & 1), qy = 2 * tile_y + (q >> 1);
    const int lane = threadIdx.x;
    const int j = qx * 8 + (lane & 7), i = qy * 8 + (lane >> 3);  // lane = the pixel's row-major index in the quadrant
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const bool inside = (i < H) && (j < W);
    // a finished pixel (outside the image, or stopped) has gate = 0: its alpha is 0 and nothing passes any more; the
    // number of unfinished pixels is a wave-uniform counter (no per-lane flag carried through the loop)
    float gate = inside ? 1.f : 0.f;
    int n_open = __popcll(__builtin_amdgcn_ballot_w64(inside));

    const int l0 = tile_offsets[tile_lin];
    const int l1 = (tile_lin == n_tiles_total - 1 && !ends_on_device) ? (int)n_isects : tile_offsets[tile_lin + 1];
    const int64_t sbase = (int64_t)l0 + 4ll * tile_lin;
    float4 *stream = rec_out ? rec_out + RS * ((int64_t)q * cap + sbase) : nullptr;
    constexpr int SS = 64 * (1 + D + E);
    float *seg_q = seg_state ? seg_state + (int64_t)q * seg_cap * SS : nullptr;

    float T = 1.f;
    int32_t cur_idx = 0;
    float pix[D], pxe[E ? E : 1];
#pragma unroll
    for (int k = 0; k < D; ++k) pix[k] = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) pxe[k] = 0.f;

    // the list is taken 64 entries at a time: lane l holds entry c0 + l; payloads are fetched two chunks ahead, the
    // records of the lanes whose entry carries this quadrant's bit one chunk ahead (for the backward's stream)
    auto load_pay = [&](int c0) -> uint32_t {
        const int idx = c0 + lane;
        return (idx < l1) ? (uint32_t)payload[idx] : 0u;
    };
    uint32_t pay_cur = load_pay(l0);
    uint32_t pay_nxt = (l0 + 64 < l1) ? load_pay(l0 + 64) : 0u;
    float4 f0, f1, f2, f3;
    auto gather = [&](uint32_t pay) {
        if ((pay >> (28 + q)) & 1u) {
            const float4 *src = packed + (int64_t)(pay & 0x0FFFFFFFu) * 4;
            f0 = src[0]; f1 = src[1]; f2 = src[2];
            if (E) f3 = src[3];
        }
    };
    gather(pay_cur);

    int cnt = 0;  // records streamed so far
    int lim = 0;  // the hot loop's end (set to 0 by the record that finishes the last open pixel)
    // (branch-free past the "does any lane pass" test: on a lone wave's critical path every instruction is a slot of its
    // own, scalar mask logic and skips included — selects instead of an exec-masked block and one loop bound instead of
    // two tests: 99 -> 92 instructions per two records, 79 -> 74 us at config #2)
    auto composite_v = [&](float mx, float my, float op, float ca, float cb, float cc, auto colour, int idx) {
        const float dx = mx - px, dy = my - py;
        const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
        // (the gate multiplies the CLAMPED value: fminf(0.999, NaN) = 0.999 must not let a record with NaN sigma commit on
        // a finished or outside lane; same instruction count, same bits for open lanes)
        const float alpha = fminf(kAlphaMax, op * __expf(-sigma)) * gate;
        // (wave masks taken from the compares themselves: a ballot of a combined flag costs two vector instructions)
        const bool ok_s = !(sigma < 0.f), ok_a = !(alpha < kAlphaMin);
        const uint64_t pm = __builtin_amdgcn_ballot_w64(ok_s) & __builtin_amdgcn_ballot_w64(ok_a);
        if (pm) {
            const float Tn = T * (1.f - alpha);
            // a record that would take T to <= 1e-4 ends the pixel and is not composited
            const bool low = Tn <= kTMin;
            const bool commit = ok_s && ok_a && !low;
            const float vis = commit ? alpha * T : 0.f;
            float c7[7];
            colour(c7);
            pix[0] += c7[0] * vis;
            if (D > 1) pix[1] += c7[1] * vis;
            if (D > 2) pix[2] += c7[2] * vis;
            if (D > 3) pix[D - 1] += c7[3] * vis;
            if (E) {
                pxe[0] += c7[4] * vis;
                if (E > 1) pxe[1] += c7[5] * vis;
                if (E > 2) pxe[E - 1] += c7[6] * vis;
            }
            cur_idx = commit ? idx : cur_idx;
            T = commit ? Tn : T;
            const uint64_t sm = __builtin_amdgcn_ballot_w64(low) & pm;
            if (sm) {
                if (ok_s && ok_a && low) gate = 0.f;
                n_open -= __popcll(sm);
                if (n_open == 0) lim = 0;
            }
        }
    };
    for (int c0 = l0; c0 < l1; c0 += 64) {
        if (n_open == 0) break;  // every pixel is opaque (or outside)
        const uint32_t pay = pay_cur;
        const bool live = (pay >> (28 + q)) & 1u;
        uint64_t m = __builtin_amdgcn_ballot_w64(live);
        if (stream && live) {
            const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
            f1.z = __int_as_float(c0 + lane);
            f1.w = __int_as_float((int)(pay & 0x0FFFFFFFu));
            float4 *dst = stream + RS * (int64_t)pos;
            dst[0] = f0; dst[1] = f1; dst[2] = f2;
            if (E) dst[3] = f3;
        }
        const float4 g0 = f0, g1 = f1, g2 = f2, g3 = f3;  // (this chunk's records, parked below)
        pay_cur = pay_nxt;
        if (c0 + 64 < l1) gather(pay_cur);
        pay_nxt = (c0 + 128 < l1) ? load_pay(c0 + 128) : 0u;
        if (!m) continue;

        {
            // the chunk's records parked in LDS in list order (structure of arrays: conflict-free 16-byte writes); every
            // step all lanes read ONE record (broadcast reads), the next record's reads issued before this one's use
            __builtin_amdgcn_s_barrier();  // (one wave: the previous chunk's reads are done)
            static_assert(E == 3, "the list index rides in the free fourth word of the normal-plane record");
            if (live) {
                const int pos = __popcll(m & ((1ull << lane) - 1ull));
                S.r0[pos] = g0; S.r1[pos] = g1; S.r2[pos] = g2;
                S.r3[pos] = make_float4(g3.x, g3.y, g3.z, __int_as_float(c0 + lane));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int n = __popcll(m);
            // Two register sets in turn (a copy per record would cost as much as the walk saves); the hot loop runs to
            // the next 64-record boundary of the stream, so the segment-state stores stay outside it.
            auto lds_rec = [&](int k, float4 &r0, float2 &r1, float4 &r2, float4 &r3) {
                r0 = S.r0[k];
                r1 = *reinterpret_cast<const float2 *>(&S.r1[k]);
                r2 = S.r2[k];
                r3 = S.r3[k];
            };
            auto step = [&](const float4 &r0, const float2 &r1, const float4 &r2, const float4 &r3) {
                composite_v(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y,
                            [&](float *c7) {
                                c7[0] = r2.x; c7[1] = r2.y; c7[2] = r2.z; c7[3] = r2.w;
                                c7[4] = r3.x; c7[5] = r3.y; c7[6] = r3.z;
                            }, __float_as_int(r3.w));
            };
            int k = 0;
            while (k < n) {
                const int run = min(n - k, 64 - (cnt & 63));
                if (n_open) {
                    float4 a0, a2, a3, b0, b2, b3;
                    float2 a1, b1;
                    lds_rec(k, a0, a1, a2, a3);
                    lim = run;
                    for (int r = 1;; r += 2) {
                        lds_rec(k + r, b0, b1, b2, b3);
                        step(a0, a1, a2, a3);
                        if (r >= lim) break;
                        lds_rec(k + r + 1, a0, a1, a2, a3);
                        step(b0, b1, b2, b3);
                        if (r + 1 >= lim) break;
                    }
                }
                // (every pixel finished inside the run: the state no longer changes, the rest only counts)
                k += run;
                cnt += run;
                if (seg_q && (cnt & 63) == 0) {
                    float *sl = seg_q + (((sbase + cnt) >> 6) + tile_lin) * SS;
                    sl[lane] = T;
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) sl[64 * (1 + kk) + lane] = pix[kk];
#pragma unroll
                    for (int kk = 0; kk < E; ++kk) sl[64 * (1 + D + kk) + lane] = pxe[kk];
                }
            }
        }
        // hand-off (see above): enough records walked, pixels still open, entries left
        if (handoff_records > 0 && cnt >= handoff_records && n_open > 0 && c0 + 64 < l1) {
            const int n_chunks = (l1 - (c0 + 64) + 63) >> 6;
            int slot0 = 0;
            if (lane == 0)
                __hip_atomic_fetch_add(&tq.header()->demand, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n_chunks <= kTailMaxChunks && tq.qmax > 0) {
                if (lane == 0) slot0 = atomicAdd(&tq.header()->counter[tq.epoch & 1u], n_chunks);
                slot0 = __builtin_amdgcn_readfirstlane(slot0);
            } else {
                slot0 = tq.qmax;
            }
            if (slot0 + n_chunks <= tq.qmax) {
                // items and state leave write-through, are drained, and only then are the positions marked filled
                for (int i = lane; i < n_chunks; i += 64) {
                    int32_t *ip = reinterpret_cast<int32_t *>(&tq.items()[slot0 + i]);
                    __hip_atomic_store(ip + 0, tile_lin | (q << 28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ip + 1, c0 + 64 + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ip + 2, slot0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ip + 3, n_chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                TailArea *a = tq.area(slot0);
                st_sc1(&a->base[0][lane], T);
#pragma unroll
                for (int kk = 0; kk < D; ++kk) st_sc1(&a->base[1 + kk][lane], pix[kk]);
#pragma unroll
                for (int kk = 0; kk < E; ++kk) st_sc1(&a->base[1 + D + kk][lane], pxe[kk]);
                st_sc1(&a->base[8][lane], __int_as_float(cur_idx));
                st_sc1(&a->base[9][lane], gate);
                if (lane == 0) __hip_atomic_store(&a->base_cnt, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int i = lane; i < n_chunks; i += 64) store_tagged(&tq.area(slot0 + i)->readyw, tq.epoch, 1u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                leave();
                return;  // the wave of this list's last chunk writes the pixels, n_rec and tile_open
            }
            // the queue is full (or the remainder too long): the claimed items that exist are voided and THIS wave walks
            // the rest of the list chunk by chunk with the second pass's arithmetic (tail_composite: same bits as if
            // the list had gone through the queue — which lists do is decided by the order of the atomics)
            if (n_chunks <= kTailMaxChunks && tq.qmax > 0) {
                for (int i = lane; i < n_chunks; i += 64)
                    if (slot0 + i < tq.qmax) store_tagged(&tq.area(slot0 + i)->readyw, tq.epoch, 2u);
            }
            float sum[D + E];
#pragma unroll
            for (int kk = 0; kk < D; ++kk) sum[kk] = pix[kk];
#pragma unroll
            for (int kk = 0; kk < E; ++kk) sum[D + kk] = pxe[kk];
            float T_in = T;
            bool open = gate != 0.f;
            // (the next chunk's entries are in flight while this chunk is composited; its records are not — their 16
            // registers across tail_composite would push the kernel past the 6-waves-per-SIMD budget into scratch)
            auto fetch = [&](int c) -> uint32_t {
                const int idx = c + lane;
                return (idx < l1) ? (uint32_t)payload[idx] : 0u;
            };
            uint32_t pc_next = fetch(c0 + 64);
            for (int c = c0 + 64; c < l1; c += 64) {
                const bool opn = open && (T_in > kTMin);
                if (__builtin_amdgcn_ballot_w64(opn) == 0ull) break;  // (the active chunks are a prefix)
                const int idx = c + lane;
                const uint32_t pc = pc_next;
                const bool lv = (pc >> (28 + q)) & 1u;
                const uint64_t mc = __builtin_amdgcn_ballot_w64(lv);
                const int n = __popcll(mc);
                __builtin_amdgcn_s_barrier();
                if (lv) {
                    const int g = (int)(pc & 0x0FFFFFFFu);
                    const float4 *src = packed + (int64_t)g * 4;
                    float4 h0 = src[0], h1 = src[1], h2 = src[2], h3 = src[3];
                    h1.z = __int_as_float(idx);
                    h1.w = __int_as_float(g);
                    const int pos = __popcll(mc & ((1ull << lane) - 1ull));
                    S.r0[pos] = h0; S.r1[pos] = h1; S.r2[pos] = h2;
                    S.r3[pos] = make_float4(h3.x, h3.y, h3.z, __int_as_float(idx));
                    if (stream) {
                        float4 *dst = stream + RS * (int64_t)(cnt + pos);
                        dst[0] = h0; dst[1] = h1; dst[2] = h2; dst[3] = h3;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (c + 64 < l1) pc_next = fetch(c + 64);
                TailChunk o;
                bool op2 = opn;
                float *slot = seg_q ? seg_q + (((sbase + ((cnt >> 6) + 1) * 64) >> 6) + tile_lin) * SS : nullptr;
                tail_composite<E>(S, n, px, py, T_in, op2, 63 - (cnt & 63), o, [&](float T_b, const float (&pl)[7]) {
                    if (slot) {
                        slot[lane] = open ? T_b : T;  // (a pixel that had stopped before this chunk keeps its T)
#pragma unroll
                        for (int kk = 0; kk < D + E; ++kk) slot[64 * (1 + kk) + lane] = pl[kk] + sum[kk];
                    }
                });
#pragma unroll
                for (int kk = 0; kk < D + E; ++kk) sum[kk] += o.pixl[kk];
                cur_idx = max(cur_idx, o.cur_idx);
                T = open ? o.T_cur : T;  // (as the finishing wave of the second pass has it)
                open = open && op2;
                {
#pragma clang fp contract(off)
                    T_in = T_in * o.p_end;
                }
                cnt += n;
            }
#pragma unroll
            for (int kk = 0; kk < D; ++kk) pix[kk] = sum[kk];
#pragma unroll
            for (int kk = 0; kk < E; ++kk) pxe[kk] = sum[D + kk];
            n_open = __popcll(__builtin_amdgcn_ballot_w64(open));
            break;
        }
    }
    if (n_rec && lane == 0) {
        n_rec[(int64_t)q * n_tiles_total + tile_lin] = cnt;
        split_backward(tile_lin, q, n_tiles_total, cnt, seg_split, bwd_queue, bwd_qcap);
    }
    if (tile_open && n_open && lane == 0) tile_open[tile_lin] = 1;
#ifdef FSGS_FWD_TRACE
    if (lane == 0 && blockIdx.x < (1u << 17)) {
        long long *tr = g_fwd_trace[blockIdx.x];
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        tr[0] = t_start; tr[1] = wall_clock64(); tr[2] = cnt; tr[3] = l1 - l0; tr[4] = hw; tr[5] = n_open;
    }
#endif

    if (backgrounds) {
#pragma unroll
        for (int k = 0; k < D; ++k) pix[k] += T * backgrounds[cam * D + k];
    }
    if (normalize_last) pix[D - 1] = pix[D - 1] / fmaxf(1.f - T, 1e-10f);  // expected depth
    if (inside) {
        const int64_t pix_id = ((int64_t)cam * H + i) * W + j;
        if (D == 4) {
            reinterpret_cast<float4 *>(render)[pix_id] = make_float4(pix[0], pix[1], pix[2], pix[D - 1]);
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) render[pix_id * D + k] = pix[k];
        }
        if (E) {
#pragma unroll
            for (int k = 0; k < E; ++k) render_extra[pix_id * E + k] = pxe[k] + T;  // background = 1
        }
        alphas[pix_id] = 1.f - T;
        last_ids[pix_id] = cur_idx;
    }
    if (max_last) {
        float mx = inside ? pix[D - 1] : 0.f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
        const int mm = __float_as_int(mx);
        if (lane == 0 && mm > 0) atomicMax(reinterpret_cast<int *>(max_last) + (bid & (kMaxCells - 1)), mm);
    }
    leave();
}


// A tail wave (see TailQueue above): queue position `item` = one 64-entry chunk of a list whose walk a base wave handed
// off.  D = 4, E = 3 (the training path's fused RGB+ED + normal-plane walk).
template <int D, int E>
__device__ __forceinline__ void
raster_fwd_tail_wave(WaveLds<E> &S, const int item, const int n_base_blocks, int64_t cap,
                     const float4 *__restrict__ packed, const int32_t *__restrict__ payload,
                     const int32_t *__restrict__ tile_offsets, int64_t n_isects,
                     const float *__restrict__ backgrounds, int W, int H, int tw, int th, int n_tiles_total,
                     float *__restrict__ render, float *__restrict__ alphas, int32_t *__restrict__ last_ids,
                     float4 *__restrict__ rec_out, int32_t *__restrict__ n_rec,
                     float *__restrict__ seg_state, int64_t seg_cap, int normalize_last,
                     float *__restrict__ render_extra, float *__restrict__ max_last, int ends_on_device,
                     int32_t *__restrict__ tile_open, const TailQueue &tq, int32_t *__restrict__ seg_split,
                     int32_t *__restrict__ bwd_queue, int bwd_qcap) {
    static_assert(D == 4 && E == 3, "the fused RGB+ED + normal-plane walk");
    constexpr int RS = 4;
    constexpr int SS = 64 * (1 + D + E);
    const int lane = threadIdx.x;
    TailHeader *hdr = tq.header();
    const uint32_t epoch = tq.epoch, par = epoch & 1u;
    int32_t *err = &hdr->error;
    // (the NEXT launch on this memory counts from zero again; it starts after this one has ended)
    if (item < 0) {
        // the watcher: waits until every walking base wave has left, tells the tail waves (64 granules, one per line)
        // and prepares the other parity's counters for the next launch on this memory
        hdr->done[par ^ 1u][lane][0] = 0;
        if (lane == 0) hdr->counter[par ^ 1u] = 0;
        for (int spin = 0;; ++spin) {
            int dn = __hip_atomic_load(&hdr->done[par][lane][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) dn += __shfl_xor(dn, dd, 64);
            if (__builtin_amdgcn_readfirstlane(dn) >= 4 * n_tiles_total) break;
            if (spin >= kTailSpinLimit) { *err = 1; break; }
            __builtin_amdgcn_s_sleep(32);
        }
        store_tagged(&hdr->all_left[lane][0], epoch, 1u);
        return;
    }
    TailArea *mine = tq.area(item);
    // wait until a base wave has filled this position — or until every base workgroup has left without doing so
    {
        uint32_t state = 0u;
        for (int spin = 0;; ++spin) {
            unsigned long long w = __hip_atomic_load(&mine->readyw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(w >> 32) == epoch) { state = (uint32_t)w; break; }
            bool left = false;
            {   // (64 lines: a few dozen waiting waves per line, one read every ~2 us each)
                const unsigned long long a = __hip_atomic_load(&hdr->all_left[item & 63][0], __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT);
                left = (uint32_t)(a >> 32) == epoch;
            }
            {
                if (left) {  // (a base wave fills its positions before it counts itself out: one last look)
                    w = __hip_atomic_load(&mine->readyw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(w >> 32) == epoch) state = (uint32_t)w;
                    break;
                }
            }
            if (spin >= kTailSpinLimit) { *err = 1; break; }
            // (back-off: a position that is not filled soon is most likely never filled)
            if (spin < 8) __builtin_amdgcn_s_sleep(24); else __builtin_amdgcn_s_sleep(64);
        }
        if (state != 1u) return;  // nothing queued here (0) or voided (2)
    }
    int4 it;
    {
        const int32_t *ip = reinterpret_cast<const int32_t *>(&tq.items()[item]);
        it.x = __hip_atomic_load(ip + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        it.y = __hip_atomic_load(ip + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        it.z = __hip_atomic_load(ip + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        it.w = __hip_atomic_load(ip + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int tile_lin = it.x & 0x0FFFFFFF, q = (it.x >> 28) & 3;
    const int c0 = it.y, slot0 = it.z, n_chunks = it.w, j = item - slot0;
    const int cam = tile_lin / (tw * th);
    const int tile_in = tile_lin - cam * tw * th;
    const int tile_y = tile_in / tw, tile_x = tile_in - tile_y * tw;
    const int qx = 2 * tile_x + (q & 1), qy = 2 * tile_y + (q >> 1);
    const int pj = qx * 8 + (lane & 7), pi = qy * 8 + (lane >> 3);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const bool inside = (pi < H) && (pj < W);
    const int l0 = tile_offsets[tile_lin];
    const int l1 = (tile_lin == n_tiles_total - 1 && !ends_on_device) ? (int)n_isects : tile_offsets[tile_lin + 1];
    const int64_t sbase = (int64_t)l0 + 4ll * tile_lin;
    float4 *stream = rec_out ? rec_out + RS * ((int64_t)q * cap + sbase) : nullptr;
    float *seg_q = seg_state ? seg_state + (int64_t)q * seg_cap * SS : nullptr;
    const TailArea *first = tq.area(slot0);

    // (0) this chunk's entries, filtered by the quadrant bit; the count is published at once (stream positions)
    const int idx = c0 + lane;
    const uint32_t pay = (idx < l1) ? (uint32_t)payload[idx] : 0u;
    const bool live = (pay >> (28 + q)) & 1u;
    const uint64_t m = __builtin_amdgcn_ballot_w64(live);
    const int n = __popcll(m);
    if (lane == 0) store_tagged(&mine->cntw, epoch, (uint32_t)n);
    float4 f0, f1, f2, f3;
    f0 = f1 = f2 = f3 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int pos = __popcll(m & ((1ull << lane) - 1ull));
    if (live) {
        const int g = (int)(pay & 0x0FFFFFFFu);
        const float4 *src = packed + (int64_t)g * 4;
        f0 = src[0]; f1 = src[1]; f2 = src[2]; f3 = src[3];
        f1.z = __int_as_float(idx);
        f1.w = __int_as_float(g);
        S.r0[pos] = f0; S.r1[pos] = f1; S.r2[pos] = f2;
        S.r3[pos] = make_float4(f3.x, f3.y, f3.z, __int_as_float(idx));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // (1) the chunk's transmittance factor per pixel: the product of (1 - alpha) over its records (no stop rule here)
    const float p_all = tail_factor<E>(S, n, px, py);
    store_tagged(&mine->P[lane], epoch, __float_as_uint(p_all));

    // (2) look-back: the handed-off state times the predecessors' factors, in chunk order; stream position likewise
    float T_in = ld_sc1(&first->base[0][lane]);
    bool open = ld_sc1(&first->base[9][lane]) != 0.f;
    int base = __hip_atomic_load(&first->base_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i0 = 0; i0 < j; i0 += 64) {
        const int i = i0 + lane;
        int c = (i < j) ? (int)wait_tagged(&tq.area(slot0 + i)->cntw, epoch, err) : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        base += c;
    }
    for (int i0 = 0; i0 < j; i0 += 8) {
        unsigned long long w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            w[u] = (i0 + u < j) ? __hip_atomic_load(&tq.area(slot0 + i0 + u)->P[lane], __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u < j) {
                uint32_t v = (uint32_t)w[u];
                if ((uint32_t)(w[u] >> 32) != epoch) v = wait_tagged(&tq.area(slot0 + i0 + u)->P[lane], epoch, err);
                {
#pragma clang fp contract(off)
                    T_in = T_in * __uint_as_float(v);
                }
            }
        }
    }
    // a pixel whose transmittance at the chunk's start is <= 1e-4 stopped inside an earlier chunk (the record that
    // took it there was not composited: the reference's rule, applied below with the same products)
    open = open && (T_in > kTMin);
    const bool active = __builtin_amdgcn_ballot_w64(open) != 0ull;

    // (3) composite the chunk from T_in; chunk-local sums; records streamed at stream position base + r
    if (active) {
        if (stream && live) {
            float4 *dst = stream + RS * (int64_t)(base + pos);
            dst[0] = f0; dst[1] = f1; dst[2] = f2; dst[3] = f3;
        }
        // the one 64-record boundary of the quadrant's stream this chunk can cross: after its record r_b; its slot gets
        // the chunk-local sums (the finishing wave adds what lies in front of this chunk)
        const int r_b = 63 - (base & 63);
        float *sl = seg_q ? seg_q + (((sbase + base + r_b + 1) >> 6) + tile_lin) * SS : nullptr;
        TailChunk o;
        tail_composite<E>(S, n, px, py, T_in, open, r_b, o, [&](float T_b, const float (&pl)[7]) {
            if (sl) {
                st_sc1(sl + lane, T_b);
#pragma unroll
                for (int k = 0; k < D + E; ++k) st_sc1(sl + 64 * (1 + k) + lane, pl[k]);
            }
        });
#pragma unroll
        for (int k = 0; k < D + E; ++k) st_sc1(&mine->part[k][lane], o.pixl[k]);
        st_sc1(&mine->part[7][lane], __int_as_float(o.cur_idx));
        st_sc1(&mine->part[8][lane], o.T_cur);
        st_sc1(&mine->part[9][lane], open ? 1.f : 0.f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (write-through stores acknowledged before the tagged word)
    }
    if (lane == 0) store_tagged(&mine->donew, epoch, active ? 2u : 1u);
    if (j != n_chunks - 1) return;

    // (4) the list's last chunk finishes the quadrant: partial sums in chunk order, segment states completed, pixels out
    int n_active = 0;  // the active chunks are a prefix (open pixels only ever close)
    for (int i0 = 0; i0 < n_chunks; i0 += 64) {
        const int i = i0 + lane;
        const bool act = (i < n_chunks) && wait_tagged(&tq.area(slot0 + i)->donew, epoch, err) == 2u;
        n_active += __popcll(__builtin_amdgcn_ballot_w64(act));
    }
    float sum[D + E];
    float T = ld_sc1(&first->base[0][lane]);
#pragma unroll
    for (int k = 0; k < D + E; ++k) sum[k] = ld_sc1(&first->base[1 + k][lane]);
    int32_t last = __float_as_int(ld_sc1(&first->base[8][lane]));
    bool op = ld_sc1(&first->base[9][lane]) != 0.f;
    int at = __hip_atomic_load(&first->base_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int kFinB = 2;  // (chunks in flight: the tail waves live inside the base waves' register budget)
    for (int i0 = 0; i0 < n_active; i0 += kFinB) {
        // several chunks' partial sums (and the boundary slots they wrote) in flight at once, combined in chunk order
        float part[kFinB][10], slot_v[kFinB][1 + D + E];
        float *slot_p[kFinB];
        int n_u[kFinB];
        int at_u = at;
#pragma unroll
        for (int u = 0; u < kFinB; ++u) {
            const bool have = i0 + u < n_active;
            const TailArea *a = tq.area(slot0 + (have ? i0 + u : i0));
            n_u[u] = have ? (int)(__hip_atomic_load(&a->cntw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFFFFFull) : 0;
#pragma unroll
            for (int k = 0; k < 10; ++k) part[u][k] = have ? ld_sc1(&a->part[k][lane]) : 0.f;
            // the boundary inside (at_u, at_u + n_u], if any: its slot holds chunk-local sums
            const int g = ((at_u >> 6) + 1) << 6;
            slot_p[u] = (have && seg_q && g <= at_u + n_u[u]) ? seg_q + (((sbase + g) >> 6) + tile_lin) * SS : nullptr;
#pragma unroll
            for (int k = 0; k < 1 + D + E; ++k) slot_v[u][k] = slot_p[u] ? ld_sc1(slot_p[u] + 64 * k + lane) : 0.f;
            at_u += n_u[u];
        }
#pragma unroll
        for (int u = 0; u < kFinB; ++u) {
            if (i0 + u < n_active) {
                if (slot_p[u]) {
                    // (the chunk knew neither the sums in front of it nor the T of pixels that had stopped before it)
                    slot_p[u][lane] = op ? slot_v[u][0] : T;
#pragma unroll
                    for (int k = 0; k < D + E; ++k) slot_p[u][64 * (1 + k) + lane] = slot_v[u][1 + k] + sum[k];
                }
#pragma unroll
                for (int k = 0; k < D + E; ++k) sum[k] += part[u][k];
                last = max(last, __float_as_int(part[u][7]));
                T = op ? part[u][8] : T;
                op = op && (part[u][9] != 0.f);
                at += n_u[u];
            }
        }
    }
    if (n_rec && lane == 0) {
        n_rec[(int64_t)q * n_tiles_total + tile_lin] = at;
        split_backward(tile_lin, q, n_tiles_total, at, seg_split, bwd_queue, bwd_qcap);
    }
    const bool any_open = __builtin_amdgcn_ballot_w64(op) != 0ull;
    if (tile_open && any_open && lane == 0) tile_open[tile_lin] = 1;
    float pix[D], pxe[E];
#pragma unroll
    for (int k = 0; k < D; ++k) pix[k] = sum[k];
#pragma unroll
    for (int k = 0; k < E; ++k) pxe[k] = sum[D + k];
    if (backgrounds) {
#pragma unroll
        for (int k = 0; k < D; ++k) pix[k] += T * backgrounds[cam * D + k];
    }
    if (normalize_last) pix[D - 1] = pix[D - 1] / fmaxf(1.f - T, 1e-10f);  // expected depth
    if (inside) {
        const int64_t pix_id = ((int64_t)cam * H + pi) * W + pj;
        reinterpret_cast<float4 *>(render)[pix_id] = make_float4(pix[0], pix[1], pix[2], pix[3]);
#pragma unroll
        for (int k = 0; k < E; ++k) render_extra[pix_id * E + k] = pxe[k] + T;  // background = 1
        alphas[pix_id] = 1.f - T;
        last_ids[pix_id] = last;
    }
    if (max_last) {
        float mx = inside ? pix[D - 1] : 0.f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
        const int mm = __float_as_int(mx);
        const unsigned bid = (unsigned)(tile_lin >> 3) * 32u + (unsigned)q * 8u + (unsigned)(tile_lin & 7);
        if (lane == 0 && mm > 0) atomicMax(reinterpret_cast<int *>(max_last) + (bid & (kMaxCells - 1)), mm);
    }
}

}  // namespace fsgs

using namespace fsgs;

// records per stream (there are four): every list entry plus 4 padding positions per tile
extern "C" int64_t fsgs_quad_stream_capacity(int C, int tile_width, int tile_height, int64_t n_isects) {
    return n_isects + 4ll * C * tile_width * tile_height;
}

// segment-state slots per stream: one per 64 stream positions plus one per tile (see the slot formula)
extern "C" int64_t fsgs_quad_seg_slots(int C, int tile_width, int tile_height, int64_t n_isects) {
    return (fsgs_quad_stream_capacity(C, tile_width, tile_height, n_isects) >> 6) +
           (int64_t)C * tile_width * tile_height + 2;
}

extern "C" int fsgs_raster_quad_max_cells(void) { return kMaxCells; }

#ifdef FSGS_FWD_TRACE
extern "C" int fsgs_debug_fwd_trace(long long *out, int n_blocks) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_trace), sizeof(long long) * 6 * n_blocks) == hipSuccess ? 0 : 1;
}
#endif

// bytes of tail scratch for a queue of `max_items` work items (fsgs_raster_fwd_quad: tail_scratch)
extern "C" int64_t fsgs_raster_fwd_tail_scratch_bytes(int max_items) {
    return max_items <= 0 ? 0 : (int64_t)kTailHeaderBytes + (16ll + kTailAreaBytes) * max_items;
}

// the error word of a tail scratch (0 = every look-back wait of every launch so far ended in time); synchronous copy
extern "C" int fsgs_raster_fwd_tail_error(const void *tail_scratch, fsgs_stream_t stream) {
    if (!tail_scratch) return FSGS_EINVAL;
    TailHeader h;
    hipError_t e = hipMemcpyAsync(&h, tail_scratch, sizeof(h), hipMemcpyDeviceToHost, as_stream(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
    return h.error != 0 ? 1 : 0;
}

extern "C" int fsgs_raster_fwd_quad(int C, int D, const float *packed, const int32_t *payload,
                                    const int32_t *isect_offsets, int64_t n_isects, const float *backgrounds,
                                    int width, int height, int tile_width, int tile_height, int normalize_last,
                                    float *render, float *alphas, int32_t *last_ids, float *records,
                                    int32_t *n_rec, float *seg_state, float *render_extra,
                                    float *max_last, int32_t *tile_open, int walk, void *tail_scratch,
                                    int64_t tail_scratch_bytes, int tail_items, int handoff_records, int handoff_rel_len,
                                    const int32_t *handoff_gate, int64_t tail_epoch, int32_t *seg_split, int32_t *bwd_queue, int bwd_queue_items,
                                    const int32_t *tile_order, fsgs_stream_t stream) {
    // n_isects < 0: -n_isects is the CAPACITY the caller sized records / seg_state for, and the lists' true end is
    // isect_offsets[C * th * tw] on the device (fsgs_bin_live_count leaves it there): no host wait for the total
    const int ends_on_device = n_isects < 0 ? 1 : 0;
    if (ends_on_device) n_isects = -n_isects;
    if (C < 0 || width < 0 || height < 0 || walk < 0 || walk > 1) return FSGS_EINVAL;
    // the one-wave walk exists for the fused RGB+ED + normal-plane launch only: asking for it elsewhere is an error, not
    // a silent four-wave launch; a hand-off needs the one-wave walk, its scratch and a positive epoch
    if (walk == FSGS_WALK_ONE_WAVE && !render_extra) return FSGS_EINVAL;
    if (handoff_records < 0 || tail_scratch_bytes < 0 || tail_items < 0) return FSGS_EINVAL;
    if (handoff_records > 0 && (walk != FSGS_WALK_ONE_WAVE || !tail_scratch)) return FSGS_EINVAL;
    if (handoff_records > 0 && tail_items > 0 && tail_epoch <= 0) return FSGS_EINVAL;
    if (C == 0 || width == 0 || height == 0) return FSGS_OK;
    if (n_isects > 0x7FFFFFF0ll) return FSGS_EINVAL;  // list indices are 32-bit (as in the reference)
    if (!isect_offsets || !render || !alphas || !last_ids || (n_isects > 0 && (!packed || !payload)))
        return FSGS_EINVAL;
    if ((records != nullptr) != (n_rec != nullptr)) return FSGS_EINVAL;
    if ((seg_split && !n_rec) || (bwd_queue && !seg_split) || bwd_queue_items < 0) return FSGS_EINVAL;
    if (seg_state && !records) return FSGS_EINVAL;
    const int64_t n_tiles = (int64_t)C * tile_width * tile_height;
    if (n_tiles >= (1ll << 26)) return FSGS_EINVAL;
    const dim3 grid((unsigned)(((n_tiles + 7) / 8) * 32));
    hipStream_t s = as_stream(stream);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    float4 *rec = reinterpret_cast<float4 *>(records);
    const int64_t seg_cap = fsgs_quad_seg_slots(C, tile_width, tile_height, n_isects);
    const int64_t cap = fsgs_quad_stream_capacity(C, tile_width, tile_height, n_isects);
#define FSGS_FWD_QUAD(DD, EE)                                                                                    \
    hipLaunchKernelGGL((raster_fwd_quad_kernel<DD, EE>), grid, dim3(256), 0, s, cap, pk, payload,                 \
                       isect_offsets, n_isects, backgrounds, width, height, tile_width, tile_height, (int)n_tiles, \
                       render, alphas, last_ids, rec, n_rec, seg_state, seg_cap, normalize_last, render_extra,     \
                       max_last, ends_on_device, tile_open, seg_split, bwd_queue, bwd_queue_items)
    if (render_extra) {
        if (D != 4) return FSGS_EINVAL;  // the fused RGB+depth + normal-plane walk
        if (walk == FSGS_WALK_ONE_WAVE) {
            // third generation: one wave per quadrant; the grid holds every quadrant twice (long lists first);
            // fourth generation (handoff_records > 0): long walks are handed to a queue and finished chunk-parallel by
            // the second launch, one wave per 64-entry chunk
            TailQueue tq = {nullptr, 0, 0u};
            if (handoff_records > 0) {
                const int64_t room = (tail_scratch_bytes - kTailHeaderBytes) / (16 + kTailAreaBytes);
                tq.mem = static_cast<uint8_t *>(tail_scratch);
                // (positions are claimed in increasing order and their workgroups dispatched in increasing order: the tail
                // waves need not all be resident at once)
                if (room < (tail_items > 0 ? tail_items : 1)) return FSGS_ESCRATCH;
                tq.qmax = tail_items > 65536 ? 65536 : tail_items;
                tq.epoch = (uint32_t)(tail_epoch & 0xFFFFFFFFll);
                if (tq.qmax > 0 && tq.epoch == 0u) return FSGS_EINVAL;
            }
            // (the tail waves = the last qmax workgroups of the launch, one per queue position)
            if (tile_order && C != 1) return FSGS_EINVAL;
            const int n_base = tile_order ? (int)grid.x : (int)grid.x * 2;
            hipLaunchKernelGGL((raster_fwd_wave_kernel<4, 3>), dim3(n_base + (tq.qmax > 0 ? tq.qmax + 1 : 0)), dim3(64), 0, s, cap,
                               pk, payload, isect_offsets, n_isects, backgrounds, width, height, tile_width, tile_height,
                               (int)n_tiles, render, alphas, last_ids, rec, n_rec, seg_state, seg_cap, normalize_last,
                               render_extra, max_last, ends_on_device, tile_open, tq.mem ? handoff_records : 0, tq, n_base,
                               handoff_rel_len < 0 ? 0 : handoff_rel_len, seg_split, bwd_queue, bwd_queue_items,
                               handoff_gate, tile_order);
        } else {
            FSGS_FWD_QUAD(4, 3);
        }
        return check_launch();
    }
    switch (D) {
        case 1: FSGS_FWD_QUAD(1, 0); break;
        case 3: FSGS_FWD_QUAD(3, 0); break;
        case 4: FSGS_FWD_QUAD(4, 0); break;
        default: return FSGS_EINVAL;
    }
#undef FSGS_FWD_QUAD
    return check_launch();
}
