// "Live lists": exact per-quadrant reach masks for the sorted tile lists (internal acceleration
// structure; the gsplat-visible lists in `meta` are untouched).
//
// Why: the sorted tile lists come from each Gaussian's square 3-sigma bounding box.  Measured on
// BASELINE config #2, only ~56 % of the (tile, Gaussian) entries can reach ANY pixel of their tile at
// alpha >= 1/255, and a surviving entry reaches ~2.1 of the tile's four 8x8 quadrants on average.
//   live_pack     one 64-byte line per (camera, Gaussian) with everything the compositing needs, so
//                 the list-order kernels gather ONE line per entry instead of five to seven 4..16-byte
//                 pieces from five arrays (optionally computing the Gaussians' camera-space normals
//                 for the extra plane in the same launch)
//   live_payload  one thread per list entry: closed-form minimum of the conic's quadratic form over
//                 each quadrant's rectangle of pixel centres vs ln(255*opacity) (+ safety margin) ->
//                 4-bit quadrant mask, stored in the top bits of the payload (flatten id < 2^28).
//                 fsgs_isect_emit_live (isect.hip) writes the same payload directly.
// The forward (raster_quad.hip) filters, gathers and composites per quadrant from these.
// Exactness: an entry is dropped for a quadrant only if no pixel centre in it can satisfy
// alpha >= 1/255, so images, last_ids and gradients equal those of the plain walk.
#include "common.h"
#include "cull.h"
#include "normal_math.h"

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
        normal_to_camera(o.n, c2w, ex);
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
