// "Live lists": an exact, massively parallel culling + compaction pass between binning and
// compositing (internal acceleration structure; the gsplat-visible lists in `meta` are untouched).
//
// Why: the sorted tile lists come from each Gaussian's square 3-sigma bounding box.  Measured on
// BASELINE config #2, only ~23 % of the (tile, Gaussian) entries can reach ANY pixel of their
// tile at alpha >= 1/255, and a surviving entry reaches 1.8 of the tile's four 8x8 quadrants on
// average.  Walking the raw lists inside the compositing kernel made it latency-bound: a serial
// per-tile chain of (index load -> dependent 44-byte gather -> barrier) per batch, mostly for
// entries that were then skipped.  On MI355X bandwidth is plentiful and latency is not, so:
//   1. live_mask    one thread per list entry (M-way parallel): closed-form minimum of the
//                   conic's quadratic form over each quadrant's rectangle of pixel centres vs
//                   ln(255*opacity) (+ safety margin) -> 4-bit quadrant mask.
//   2. scan4        exclusive prefix sums of the four mask bits over the whole list (one pass over
//                   M bytes, four counters at once).
//   3. live_compact one thread per entry: for every quadrant it reaches, write a self-contained
//                   48-byte record (xy, opacity, conic, list index, Gaussian id, colour) at its
//                   rank in that quadrant's stream.  Depth order is preserved (stable).
// The compositing kernels then run ONE wave per 8x8 quadrant over a contiguous, coalesced record
// stream: no dependent gathers, no barriers, no wasted entries.  A quadrant's list is
// stream q, range [pos4[offsets[tile]].q, pos4[offsets[tile+1]].q).
// Exactness: an entry is dropped for a quadrant only if no pixel centre in it can satisfy
// alpha >= 1/255, so images, last_ids and gradients equal those of the plain walk.
#include "common.h"
#include "cull.h"
#include "normal_math.h"
#include "scan.h"

namespace fsgs {

// One 64-byte line per (camera, Gaussian): {x, y, opacity, conic.a} {conic.b, conic.c, 0, 0}
// {colour, zero padded} {extra xyz, 0}.  The list-order kernels below then gather ONE line per entry
// instead of five to seven 4..16-byte pieces from five arrays (Gaussians sit in arbitrary storage
// order, so every piece is its own sector fetch: PMC showed 268 MB fetched by live_compact).
template <int D>
__global__ void __launch_bounds__(256)
live_pack_kernel(int64_t total, const float *__restrict__ means2d, const float *__restrict__ conics,
                 const float *__restrict__ colors, const float *__restrict__ opacities,
                 const float *__restrict__ extra, float4 *__restrict__ packed, float *__restrict__ zero_cells,
                 int n_zero, const float *__restrict__ quats, const float *__restrict__ log_scales,
                 const float *__restrict__ means, const float *__restrict__ c2w,
                 float *__restrict__ normals_world) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && zero_cells)  // e.g. the forward's running image maxima
        for (int k = threadIdx.x; k < n_zero; k += blockDim.x) zero_cells[k] = 0.f;
    if (g >= total) return;
    float ex[3] = {0.f, 0.f, 0.f};
    bool has_extra = extra != nullptr;
    if (quats) {
        // the extra plane is the Gaussian's camera-space normal, computed here (dn_model.py:618-636)
        // instead of by a separate launch; normals_world is the side output get_outputs returns
        const float4 q = reinterpret_cast<const float4 *>(quats)[g];
        const float ls[3] = {log_scales[g * 3], log_scales[g * 3 + 1], log_scales[g * 3 + 2]};
        const float mean[3] = {means[g * 3], means[g * 3 + 1], means[g * 3 + 2]};
        NormalCtx o;
        normal_forward(q, ls, mean, c2w, o);
#pragma unroll
        for (int k = 0; k < 3; ++k) normals_world[g * 3 + k] = o.n[k];
#pragma unroll
        for (int j = 0; j < 3; ++j) ex[j] = o.n[0] * c2w[0 * 4 + j] + o.n[1] * c2w[1 * 4 + j] + o.n[2] * c2w[2 * 4 + j];
        has_extra = true;
    } else if (extra) {
        ex[0] = extra[g * 3 + 0]; ex[1] = extra[g * 3 + 1]; ex[2] = extra[g * 3 + 2];
    }
    const float2 xy = reinterpret_cast<const float2 *>(means2d)[g];
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (D == 4) {
        c = reinterpret_cast<const float4 *>(colors)[g];
    } else {
        c.x = colors[g * D + 0];
        if (D > 1) c.y = colors[g * D + 1];
        if (D > 2) c.z = colors[g * D + 2];
    }
    packed[g * 4 + 0] = make_float4(xy.x, xy.y, opacities[g], conics[g * 3 + 0]);
    packed[g * 4 + 1] = make_float4(conics[g * 3 + 1], conics[g * 3 + 2], 0.f, 0.f);
    packed[g * 4 + 2] = c;
    packed[g * 4 + 3] = has_extra ? make_float4(ex[0], ex[1], ex[2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(256)
live_mask_packed_kernel(int64_t M, const int64_t *__restrict__ isect_ids, const int32_t *__restrict__ flatten_ids,
                        const float4 *__restrict__ packed, int tw, int tile_bits, uint8_t *__restrict__ mask8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    if (i == M) { mask8[i] = 0; return; }
    const int tile = (int)((isect_ids[i] >> 32) & ((1ll << tile_bits) - 1));
    const int ty = tile / tw, tx = tile - ty * tw;
    const int64_t g = flatten_ids[i];
    const float4 r0 = packed[g * 4 + 0], r1 = packed[g * 4 + 1];
    mask8[i] = (uint8_t)quadrant_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, (float)(tx * 16), (float)(ty * 16));
}

__global__ void __launch_bounds__(256)
live_compact_packed_kernel(int64_t M, int64_t cap, int rs, const uint8_t *__restrict__ mask8,
                           const int4 *__restrict__ pos4, const int32_t *__restrict__ flatten_ids,
                           const float4 *__restrict__ packed, float4 *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const unsigned m = mask8[i];
    if (!m) return;
    const int4 p = pos4[i];
    const int32_t g = flatten_ids[i];
    const float4 *src = packed + (int64_t)g * 4;
    const float4 r0 = src[0];
    float4 r1 = src[1];
    const float4 r2 = src[2];
    r1.z = __int_as_float((int)i);
    r1.w = __int_as_float(g);
    float4 r3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rs == 4) r3 = src[3];
    const int pq[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (m & (1u << q)) {
            float4 *dst = rec + rs * ((int64_t)q * cap + pq[q]);
            dst[0] = r0; dst[1] = r1; dst[2] = r2;
            if (rs == 4) dst[3] = r3;
        }
    }
}

__global__ void __launch_bounds__(256)
live_mask_kernel(int64_t M, const int64_t *__restrict__ isect_ids, const int32_t *__restrict__ flatten_ids,
                 const float *__restrict__ means2d, const float *__restrict__ conics,
                 const float *__restrict__ opacities, int tw, int tile_bits, uint8_t *__restrict__ mask8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    if (i == M) { mask8[i] = 0; return; }  // sentinel so that pos4[M] holds the totals
    const int tile = (int)((isect_ids[i] >> 32) & ((1ll << tile_bits) - 1));
    const int ty = tile / tw, tx = tile - ty * tw;
    const int32_t g = flatten_ids[i];
    const float2 m = reinterpret_cast<const float2 *>(means2d)[g];
    const float a = conics[(int64_t)g * 3 + 0], b = conics[(int64_t)g * 3 + 1], c = conics[(int64_t)g * 3 + 2];
    mask8[i] = (uint8_t)quadrant_mask(m.x, m.y, opacities[g], a, b, c, (float)(tx * 16), (float)(ty * 16));
}

// ---- four prefix sums at once over the mask bits ------------------------------------------
constexpr int kS4Block = 256, kS4Items = 8, kS4Tile = kS4Block * kS4Items;

__device__ __forceinline__ int4 bits4(unsigned m) {
    return make_int4(m & 1u, (m >> 1) & 1u, (m >> 2) & 1u, (m >> 3) & 1u);
}
__device__ __forceinline__ int4 add4(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ int4 wave_incl_scan4(int4 v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int ox = __shfl_up(v.x, d, 64), oy = __shfl_up(v.y, d, 64);
        const int oz = __shfl_up(v.z, d, 64), ow = __shfl_up(v.w, d, 64);
        if (lane >= d) v = add4(v, make_int4(ox, oy, oz, ow));
    }
    return v;
}

// exclusive scan of one int4 per thread across the block; *total = block sum
__device__ __forceinline__ int4 block_excl_scan4(int4 v, int4 *total, int4 *lds4) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int4 inc = wave_incl_scan4(v);
    if (lane == 63) lds4[w] = inc;
    __syncthreads();
    int4 base = make_int4(0, 0, 0, 0), tot = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < kS4Block / 64; ++k) {
        if (k < w) base = add4(base, lds4[k]);
        tot = add4(tot, lds4[k]);
    }
    *total = tot;
    __syncthreads();
    return make_int4(base.x + inc.x - v.x, base.y + inc.y - v.y, base.z + inc.z - v.z, base.w + inc.w - v.w);
}

__global__ void __launch_bounds__(kS4Block)
scan4_reduce_kernel(int64_t n, const uint8_t *__restrict__ mask8, int4 *__restrict__ block_sums) {
    __shared__ int4 lds4[4];
    const int64_t base = (int64_t)blockIdx.x * kS4Tile;
    int4 s = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < kS4Items; ++k) {
        const int64_t i = base + k * kS4Block + threadIdx.x;
        if (i < n) s = add4(s, bits4(mask8[i]));
    }
    int4 tot;
    block_excl_scan4(s, &tot, lds4);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kS4Block)
scan4_apply_kernel(int64_t n, const uint8_t *__restrict__ mask8, const int4 *__restrict__ block_sums,
                   int4 *__restrict__ pos4) {
    __shared__ int4 lds4[4];
    const int64_t base = (int64_t)blockIdx.x * kS4Tile + (int64_t)threadIdx.x * kS4Items;
    unsigned m[kS4Items];
    int4 s = make_int4(0, 0, 0, 0);
    if (base + kS4Items <= n) {
        const uint2 raw = *reinterpret_cast<const uint2 *>(mask8 + base);  // 8 masks, 8-byte aligned
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = (raw.x >> (8 * k)) & 0xFFu;
            m[4 + k] = (raw.y >> (8 * k)) & 0xFFu;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kS4Items; ++k) m[k] = (base + k < n) ? mask8[base + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kS4Items; ++k) s = add4(s, bits4(m[k]));
    int4 before = make_int4(0, 0, 0, 0);  // totals of all earlier workgroups, summed here (no scan launch)
    for (int64_t b = threadIdx.x; b < (int64_t)blockIdx.x; b += kS4Block) before = add4(before, block_sums[b]);
    int4 btot;
    block_excl_scan4(before, &btot, lds4);
    int4 tot;
    int4 run = add4(block_excl_scan4(s, &tot, lds4), btot);
#pragma unroll
    for (int k = 0; k < kS4Items; ++k) {
        if (base + k < n) pos4[base + k] = run;
        run = add4(run, bits4(m[k]));
    }
}

// Records (RS float4 each, RS = 3 or 4): [0] = {x, y, opacity, conic.a}; [1] = {conic.b, conic.c,
// bits(list index), bits(Gaussian id)}; [2] = colour (D floats, zero padded); [3] (only when an
// `extra` per-Gaussian 3-vector rides along, e.g. the normal plane) = {e0, e1, e2, 0}.
// Stream q starts at record q*cap.
template <int D>
__global__ void __launch_bounds__(256)
live_compact_kernel(int64_t M, int64_t cap, const uint8_t *__restrict__ mask8, const int4 *__restrict__ pos4,
                    const int32_t *__restrict__ flatten_ids, const float *__restrict__ means2d,
                    const float *__restrict__ conics, const float *__restrict__ colors,
                    const float *__restrict__ opacities, const float *__restrict__ extra,
                    float4 *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const unsigned m = mask8[i];
    if (!m) return;
    const int4 p = pos4[i];
    const int32_t g = flatten_ids[i];
    const float2 xy = reinterpret_cast<const float2 *>(means2d)[g];
    const float a = conics[(int64_t)g * 3 + 0], b = conics[(int64_t)g * 3 + 1], c = conics[(int64_t)g * 3 + 2];
    const float4 r0 = make_float4(xy.x, xy.y, opacities[g], a);
    const float4 r1 = make_float4(b, c, __int_as_float((int)i), __int_as_float(g));
    float4 r2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (D == 4) {
        r2 = reinterpret_cast<const float4 *>(colors)[g];
    } else {
        r2.x = colors[(int64_t)g * D + 0];
        if (D > 1) r2.y = colors[(int64_t)g * D + 1];
        if (D > 2) r2.z = colors[(int64_t)g * D + 2];
    }
    const int pq[4] = {p.x, p.y, p.z, p.w};
    const int RS = extra ? 4 : 3;
    float4 r3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (extra) r3 = make_float4(extra[(int64_t)g * 3 + 0], extra[(int64_t)g * 3 + 1], extra[(int64_t)g * 3 + 2], 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (m & (1u << q)) {
            float4 *dst = rec + RS * ((int64_t)q * cap + pq[q]);
            dst[0] = r0; dst[1] = r1; dst[2] = r2;
            if (extra) dst[3] = r3;
        }
    }
}

// (quadrant mask << 28 | flatten id) for lists that arrive without masks (the gsplat-visible lists of
// fsgs_isect_emit): what fsgs_isect_emit_live writes directly.
__global__ void __launch_bounds__(256)
live_payload_kernel(int64_t M, const int64_t *__restrict__ isect_ids, const int32_t *__restrict__ flatten_ids,
                    const float4 *__restrict__ packed, int tw, int tile_bits, int32_t *__restrict__ payload) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int tile = (int)((isect_ids[i] >> 32) & ((1ll << tile_bits) - 1));
    const int ty = tile / tw, tx = tile - ty * tw;
    const int64_t g = flatten_ids[i];
    const float4 r0 = packed[g * 4 + 0], r1 = packed[g * 4 + 1];
    const unsigned m = quadrant_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, (float)(tx * 16), (float)(ty * 16));
    payload[i] = (int32_t)((m << 28) | ((uint32_t)g & 0x0FFFFFFFu));
}

// Live emission (isect.hip) carries the quadrant mask in the payload's top 4 bits: split it into
// the mask array and clean flatten ids (one pass over the ~M/4 live entries).
__global__ void __launch_bounds__(256)
payload_split_kernel(int64_t M, int32_t *__restrict__ payload_ids, uint8_t *__restrict__ mask8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    if (i == M) { mask8[i] = 0; return; }
    const uint32_t v = (uint32_t)payload_ids[i];
    mask8[i] = (uint8_t)(v >> 28);
    payload_ids[i] = (int32_t)(v & 0x0FFFFFFFu);
}

inline int64_t scan4_blocks(int64_t n) { return (n + kS4Tile - 1) / kS4Tile; }

}  // namespace fsgs

using namespace fsgs;

static int live_pack_impl(int D, int64_t n_gauss_total, const float *means2d, const float *conics,
                          const float *colors, const float *opacities, const float *extra, float *packed,
                          float *zero_cells, int n_zero, const float *quats, const float *log_scales,
                          const float *means, const float *c2w, float *normals_world, fsgs_stream_t stream) {
    if (n_gauss_total < 0 || (D != 1 && D != 3 && D != 4)) return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
    if (n_gauss_total == 0) {
        if (zero_cells && n_zero > 0 && hipMemsetAsync(zero_cells, 0, sizeof(float) * n_zero, s) != hipSuccess)
            return FSGS_ELAUNCH;
        return FSGS_OK;
    }
    if (!means2d || !conics || !colors || !opacities || !packed) return FSGS_EINVAL;
    float4 *pk = reinterpret_cast<float4 *>(packed);
    const dim3 gp(ceil_div(n_gauss_total, 256));
    if (D == 4)
        hipLaunchKernelGGL((live_pack_kernel<4>), gp, dim3(256), 0, s, n_gauss_total, means2d, conics, colors,
                           opacities, extra, pk, zero_cells, n_zero, quats, log_scales, means, c2w, normals_world);
    else if (D == 3)
        hipLaunchKernelGGL((live_pack_kernel<3>), gp, dim3(256), 0, s, n_gauss_total, means2d, conics, colors,
                           opacities, extra, pk, zero_cells, n_zero, quats, log_scales, means, c2w, normals_world);
    else
        hipLaunchKernelGGL((live_pack_kernel<1>), gp, dim3(256), 0, s, n_gauss_total, means2d, conics, colors,
                           opacities, extra, pk, zero_cells, n_zero, quats, log_scales, means, c2w, normals_world);
    return check_launch();
}

extern "C" int fsgs_live_pack(int D, int64_t n_gauss_total, const float *means2d, const float *conics,
                              const float *colors, const float *opacities, const float *extra, float *packed,
                              float *zero_cells, int n_zero, fsgs_stream_t stream) {
    return live_pack_impl(D, n_gauss_total, means2d, conics, colors, opacities, extra, packed, zero_cells, n_zero,
                          nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

// fsgs_live_pack with the extra plane = the Gaussians' camera-space normals, computed in the same launch
// (what fsgs_normals_fwd would write: normals_world [N,3] is still returned; single camera)
extern "C" int fsgs_live_pack_normals(int64_t N, const float *means2d, const float *conics, const float *colors,
                                      const float *opacities, const float *quats, const float *log_scales,
                                      const float *means, const float *c2w, float *packed, float *normals_world,
                                      float *zero_cells, int n_zero, fsgs_stream_t stream) {
    if (N > 0 && (!quats || !log_scales || !means || !c2w || !normals_world)) return FSGS_EINVAL;
    return live_pack_impl(4, N, means2d, conics, colors, opacities, nullptr, packed, zero_cells, n_zero, quats,
                          log_scales, means, c2w, normals_world, stream);
}

extern "C" int fsgs_live_payload(const int64_t *isect_ids, const int32_t *flatten_ids, int64_t n_isects,
                                 const float *packed, int64_t n_gauss_total, int tile_width, int tile_bits,
                                 int32_t *payload, fsgs_stream_t stream) {
    if (n_isects < 0 || n_gauss_total < 0 || n_gauss_total >= (1ll << 28)) return FSGS_EINVAL;
    if (n_isects == 0) return FSGS_OK;
    if (!isect_ids || !flatten_ids || !packed || !payload) return FSGS_EINVAL;
    hipLaunchKernelGGL(live_payload_kernel, dim3(ceil_div(n_isects, 256)), dim3(256), 0, as_stream(stream), n_isects,
                       isect_ids, flatten_ids, reinterpret_cast<const float4 *>(packed), tile_width, tile_bits,
                       payload);
    return check_launch();
}

extern "C" size_t fsgs_live_scratch_bytes(int64_t n_isects) {
    return (size_t)(scan4_blocks(n_isects + 1) + 1) * sizeof(int4);
}

extern "C" int fsgs_live_prepare(int D, const float *means2d, const float *conics, const float *colors,
                                 const float *opacities, const float *extra, int tile_width, int tile_bits,
                                 const int64_t *isect_ids, const int32_t *flatten_ids, int64_t n_isects,
                                 int64_t n_gauss_total, float *packed, int masks_in_payload, uint8_t *mask8,
                                 int32_t *pos4, float *records, void *scratch, size_t scratch_bytes,
                                 fsgs_stream_t stream) {
    if (n_isects < 0 || (D != 1 && D != 3 && D != 4)) return FSGS_EINVAL;
    if (!mask8 || !pos4) return FSGS_EINVAL;
    if (n_isects > 0x7FFFFFF0ll) return FSGS_EINVAL;  // list indices are 32-bit (as in the reference)
    hipStream_t s = as_stream(stream);
    const int64_t n = n_isects + 1;
    if (scratch_bytes < fsgs_live_scratch_bytes(n_isects) || !scratch) return FSGS_ESCRATCH;
    if (n_isects > 0 && (!means2d || !conics || !colors || !opacities || !isect_ids || !flatten_ids || !records))
        return FSGS_EINVAL;
    int4 *block_sums = reinterpret_cast<int4 *>(scratch);
    float4 *pk = reinterpret_cast<float4 *>(packed);
    const bool use_packed = packed != nullptr && n_gauss_total > 0 && n_isects > 0;
    if (masks_in_payload) {
        // flatten_ids arrive as (mask << 28 | id) from fsgs_isect_emit_live: nothing to gather for the mask
        hipLaunchKernelGGL(payload_split_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n_isects,
                           const_cast<int32_t *>(flatten_ids), mask8);
    }
    if (use_packed) {
        const dim3 gp(ceil_div(n_gauss_total, 256));
        if (D == 4)
            hipLaunchKernelGGL((live_pack_kernel<4>), gp, dim3(256), 0, s, n_gauss_total, means2d, conics, colors,
                               opacities, extra, pk, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
        else if (D == 3)
            hipLaunchKernelGGL((live_pack_kernel<3>), gp, dim3(256), 0, s, n_gauss_total, means2d, conics, colors,
                               opacities, extra, pk, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
        else
            hipLaunchKernelGGL((live_pack_kernel<1>), gp, dim3(256), 0, s, n_gauss_total, means2d, conics, colors,
                               opacities, extra, pk, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (!masks_in_payload)
            hipLaunchKernelGGL(live_mask_packed_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n_isects,
                               isect_ids, flatten_ids, pk, tile_width, tile_bits, mask8);
    } else if (!masks_in_payload) {
        hipLaunchKernelGGL(live_mask_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n_isects, isect_ids,
                           flatten_ids, means2d, conics, opacities, tile_width, tile_bits, mask8);
    }
    const int64_t nb = scan4_blocks(n);
    hipLaunchKernelGGL(scan4_reduce_kernel, dim3((unsigned)nb), dim3(kS4Block), 0, s, n, mask8, block_sums);
    hipLaunchKernelGGL(scan4_apply_kernel, dim3((unsigned)nb), dim3(kS4Block), 0, s, n, mask8, block_sums,
                       reinterpret_cast<int4 *>(pos4));
    if (n_isects > 0 && use_packed) {
        hipLaunchKernelGGL(live_compact_packed_kernel, dim3(ceil_div(n_isects, 256)), dim3(256), 0, s, n_isects,
                           n_isects, extra ? 4 : 3, mask8, reinterpret_cast<const int4 *>(pos4), flatten_ids, pk,
                           reinterpret_cast<float4 *>(records));
    } else if (n_isects > 0) {
        const dim3 grid(ceil_div(n_isects, 256));
        float4 *rec = reinterpret_cast<float4 *>(records);
        const int4 *p4 = reinterpret_cast<const int4 *>(pos4);
        if (D == 4)
            hipLaunchKernelGGL((live_compact_kernel<4>), grid, dim3(256), 0, s, n_isects, n_isects, mask8, p4,
                               flatten_ids, means2d, conics, colors, opacities, extra, rec);
        else if (D == 3)
            hipLaunchKernelGGL((live_compact_kernel<3>), grid, dim3(256), 0, s, n_isects, n_isects, mask8, p4,
                               flatten_ids, means2d, conics, colors, opacities, extra, rec);
        else
            hipLaunchKernelGGL((live_compact_kernel<1>), grid, dim3(256), 0, s, n_isects, n_isects, mask8, p4,
                               flatten_ids, means2d, conics, colors, opacities, extra, rec);
    }
    return check_launch();
}
