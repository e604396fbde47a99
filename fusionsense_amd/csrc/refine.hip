// a-13 / a-14: one refinement of /root/reference/dn_splatter/dn_model.py:326-451 (or one cull of the pruning callbacks,
// :1156-1302) in two launches with one host read between them (include/fsgs.h, "a-13 / a-14 in two launches"):
//   refine_mark_kernel  every test of refinement_after / cull_gaussians / hull_pruning / touch_pruning that is a function
//                       of the row alone -> five bits per row, per-workgroup counts, and — by the workgroup that arrives
//                       last — their exclusive offsets and the five totals, the latter into mapped host memory;
//   refine_move_kernel  the one-pass kernel of SURVEY.md section 7 step 7: every kept row of every parameter, Adam moment and
//                       statistics column lands at its final place, split children are sampled and written where
//                       torch.cat([old, repeat(samps), dups]) + cull would have left them, new rows get zero moments.
// HBM-bound row movement (177 floats per Gaussian read once, written once); the counts are ballots, the offsets one
// workgroup's scan of N/256 numbers.  No torch, no hipCUB.
#include "common.h"

namespace fsgs {

constexpr int kRefRows = 256;      // rows per workgroup, both launches
constexpr int kRefCounts = 5;      // KEEP, SPLIT, DUP, SPLIT & CHILD_KEEP, DUP & DUP_KEEP
constexpr int kRefBookHead = 8;    // book[0] = arrival ticket; the offsets start at book[8]
constexpr int kRefMaxCols = 24;
enum : unsigned { RF_KEEP = 1, RF_SPLIT = 2, RF_DUP = 4, RF_CHILD_KEEP = 8, RF_DUP_KEEP = 16 };
enum : int { COL_COPY = 0, COL_ZERO = 1, COL_MEANS = 2, COL_SCALES = 3 };

// x / 1.6 as torch divides a tensor by a Python scalar: multiplication by the fp32 reciprocal (split_gaussians'
// ``torch.log(torch.exp(scales) / size_fac)``, nerfstudio splatfacto; restated in oracle/splatfacto_ref.py)
__device__ __forceinline__ float shrink_scale(float e) { return e * (1.0f / 1.6f); }

__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
}

// PARTS threads per row (1 or 4; a row's PARTS lanes are neighbours in one wave): with the hull's grid the search of the
// 27 cells around a row is 9 runs of cells (x-neighbours are contiguous), whose points the row's lanes share out and meet
// over by a minimum.  Everything else is computed by all the row's lanes alike; lane 0 of the row writes the flag and is
// the only one counted.  Measured at config #3's shapes (75 000 rows, 20 000 of them near a hull of 20 000 points in a
// ball of radius 0.1, 29 points per cell; tools/bench_refine_mark.py, launch + wait as the host sees them): 34 us without
// the search, 105 / 75 / 71 us with it at 1 / 2 / 4 lanes per row; the exhaustive search it replaces
// (fsgs_nearest_point_words) is 380 us.
template <int PARTS>
__global__ void __launch_bounds__(256 * PARTS)
refine_mark_kernel(int64_t N, int nb, fsgs_refine_rules R, const float *__restrict__ xgn, const float *__restrict__ vis,
                   const float *__restrict__ max2d, const float *__restrict__ means, const float *__restrict__ log_scales,
                   const float *__restrict__ opac, const uint8_t *__restrict__ add_mask, const uint8_t *__restrict__ extra,
                   const unsigned long long *__restrict__ hull_best, const float4 *__restrict__ hull_points,
                   const int32_t *__restrict__ hull_cells, const float *__restrict__ boxes,
                   uint8_t *__restrict__ flags, int32_t *__restrict__ book, int32_t *__restrict__ totals_mapped) {
    __shared__ int wcnt[4 * PARTS][kRefCounts];
    __shared__ int last_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int part = tid % PARTS;
    const int64_t n = (int64_t)blockIdx.x * kRefRows + tid / PARTS;
    unsigned f = 0;
    if (n < N) {
        const float e0 = expf(log_scales[n * 3]), e1 = expf(log_scales[n * 3 + 1]), e2 = expf(log_scales[n * 3 + 2]);
        const float smax = fmaxf(fmaxf(e0, e1), e2);
        // the children's scales, log(exp(s) / 1.6), looked at the way cull_gaussians does: exp(.).max
        const float cmax = fmaxf(fmaxf(expf(logf(shrink_scale(e0))), expf(logf(shrink_scale(e1)))), expf(logf(shrink_scale(e2))));
        const bool low_alpha = 1.f / (1.f + expf(-opac[n])) < R.cull_alpha_thresh;
        const bool add = add_mask && add_mask[n];
        const float m2 = max2d ? max2d[n] : 0.f;
        bool culled = low_alpha || (extra && extra[n]);
        bool split = false, dup = false;
        if (R.densify) {
            const float avg = (xgn[n] / vis[n]) * 0.5f * R.max_hw;
            const bool high = avg > R.densify_grad_thresh;
            split = smax > R.densify_size_thresh;
            if (R.check_screen && max2d) split = split || m2 > R.split_screen_size;
            split = split && high && !add;
            const float after = split ? cmax : smax;  // (the parent's scales once split_gaussians has divided them)
            dup = after <= R.densify_size_thresh && high && !add;
            culled = culled || split;
            const bool child_big = R.check_big && cmax > R.cull_scale_thresh;
            const bool dup_big = R.check_big && after > R.cull_scale_thresh;
            if (split) f |= RF_SPLIT | ((low_alpha || child_big) ? 0u : RF_CHILD_KEEP);
            if (dup) f |= RF_DUP | ((low_alpha || dup_big) ? 0u : RF_DUP_KEEP);
        }
        if (R.check_big) {
            bool big = smax > R.cull_scale_thresh;
            if (R.check_screen && max2d) big = big || m2 > R.cull_screen_size;
            culled = culled || big;
        }
        if ((hull_best || hull_points || R.n_boxes > 0) && !add) {
            const float x = means[n * 3], y = means[n * 3 + 1], z = means[n * 3 + 2];
            if (hull_best || hull_points) {
                const float dx = x - R.hull_center[0], dy = y - R.hull_center[1], dz = z - R.hull_center[2];
                const bool close = sqrtf(dx * dx + dy * dy + dz * dz) <= R.hull_close;
                float d = INFINITY;
                if (hull_best) {
                    const unsigned long long wd = hull_best[n];
                    if (wd != ~0ull) d = sqrtf(__uint_as_float((unsigned)(wd >> 32)));
                } else if (close) {
                    // the nearest hull point among the 27 cells around the row (exact squared differences, as
                    // nearest_point_kernel forms them): all that can lie within hull_hi
                    const float gx = floorf((x - R.grid_origin[0]) * R.grid_inv_cell), gy = floorf((y - R.grid_origin[1]) * R.grid_inv_cell),
                                gz = floorf((z - R.grid_origin[2]) * R.grid_inv_cell);
                    const int cx = (int)fminf(fmaxf(gx, -2.f), (float)R.grid_dims[0] + 1.f), cy = (int)fminf(fmaxf(gy, -2.f), (float)R.grid_dims[1] + 1.f),
                              cz = (int)fminf(fmaxf(gz, -2.f), (float)R.grid_dims[2] + 1.f);
                    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, R.grid_dims[0] - 1);
                    float best = INFINITY;
                    if (x0 <= x1) {
                        // the 9 runs (dz, dy) of x-contiguous cells; the row's lanes walk every run TOGETHER, lane p taking
                        // the points p, p + PARTS, ...: neighbouring lanes read neighbouring 16-byte points (the launch is
                        // bound by the texture path's rate for scattered addresses, not by arithmetic: one lane per row
                        // reading its own point each time was 64 cache lines per wave instruction)
                        for (int zz = max(cz - 1, 0); zz <= min(cz + 1, R.grid_dims[2] - 1); ++zz)
                            for (int yy = max(cy - 1, 0); yy <= min(cy + 1, R.grid_dims[1] - 1); ++yy) {
                                const int id0 = (zz * R.grid_dims[1] + yy) * R.grid_dims[0];
                                const int j1 = hull_cells[id0 + x1 + 1];
                                // (four loads in flight per lane — a lane waits a memory latency per trip otherwise; an
                                // index beyond the run reads the run's last point again, which a minimum does not notice)
                                for (int j = hull_cells[id0 + x0] + part; j < j1; j += 4 * PARTS) {
                                    float4 p[4];
#pragma unroll
                                    for (int u = 0; u < 4; ++u) p[u] = hull_points[min(j + u * PARTS, j1 - 1)];
#pragma unroll
                                    for (int u = 0; u < 4; ++u) {
                                        const float ex = x - p[u].x, ey = y - p[u].y, ez = z - p[u].z;
                                        best = fminf(best, fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
                                    }
                                }
                            }
                    }
                    d = best;  // (squared; the row's lanes meet below)
                }
                if (!hull_best) {
                    // (close and add are the same on all lanes of a row: the exchange is uniform within the row's lanes)
                    if (PARTS > 1) {
#pragma unroll
                        for (int m = 1; m < PARTS; m <<= 1) d = fminf(d, __shfl_xor(d, m, 64));
                    }
                    d = sqrtf(d);
                }
                culled = culled || (close && d > R.hull_lo && d <= R.hull_hi);
            }
            for (int b = 0; b < R.n_boxes; ++b) {  // (points_in_boxes_kernel's test, csrc/densify.hip)
                const float *B = boxes + 18 * b;
                const float dx = x - B[0], dy = y - B[1], dz = z - B[2];
                bool in = true;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float r = fmaf(dz, B[3 + 3 * k + 2], fmaf(dy, B[3 + 3 * k + 1], dx * B[3 + 3 * k + 0]));
                    in = in && (r >= B[12 + k]) && (r <= B[15 + k]);
                }
                culled = culled || in;
            }
        }
        if (!culled) f |= RF_KEEP;
        if (part == 0) flags[n] = (uint8_t)f;
    }
    if (part != 0) f = 0;  // (one lane per row is counted)
    const unsigned want[kRefCounts] = {RF_KEEP, RF_SPLIT, RF_DUP, RF_SPLIT | RF_CHILD_KEEP, RF_DUP | RF_DUP_KEEP};
#pragma unroll
    for (int k = 0; k < kRefCounts; ++k) {
        const int c = __popcll(__ballot((f & want[k]) == want[k]));
        if (lane == 0) wcnt[w][k] = c;
    }
    __syncthreads();
    if (tid < kRefCounts) {  // (write-through: the workgroup that arrives last reads every workgroup's counts in this launch)
        int c = 0;
#pragma unroll
        for (int j = 0; j < 4 * PARTS; ++j) c += wcnt[j][tid];
        __hip_atomic_store(&book[kRefBookHead + (int64_t)tid * nb + blockIdx.x], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (no fences: an agent-scope fence per wave writes back and invalidates L2 — measured here as 18 us per doubling of
    // the workgroup.  The counts are stored write-through, drained, then ticketed; the last workgroup reads them with
    // agent-scope loads: the idiom of scan_rows_sh_pack_kernel, csrc/sh.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last_s = __hip_atomic_fetch_add(&book[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1;
    __syncthreads();
    if (!last_s) return;
    // exclusive offsets of the five count sequences, in place; totals to the host
    __shared__ int wsum[4];
    const int per = (nb + 255) / 256;  // (the first 256 threads scan; the others only keep the barriers company)
    const int i0 = tid < 256 ? min(tid * per, nb) : nb, i1 = min(i0 + per, nb);
    for (int k = 0; k < kRefCounts; ++k) {
        int32_t *seq = book + kRefBookHead + (int64_t)k * nb;
        int mine = 0;
        for (int i = i0; i < i1; ++i) mine += __hip_atomic_load(&seq[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        __syncthreads();  // (wsum of the previous sequence has been read)
        if (lane == 63 && w < 4) wsum[w] = inc;
        __syncthreads();
        int run = inc - mine;
        for (int j = 0; j < min(w, 4); ++j) run += wsum[j];
        for (int i = i0; i < i1; ++i) {
            const int c = __hip_atomic_load(&seq[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seq[i] = run;
            run += c;
        }
        if (tid == 255) {
            const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            book[1 + k] = total;
            totals_mapped[k] = total;
        }
    }
    if (tid == 255) {
        book[0] = 0;  // the ticket, for the next call
        __threadfence_system();
        __hip_atomic_store(&totals_mapped[7], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // "the totals have landed"
    }
}

struct RefineMove {
    const float *src[kRefMaxCols];
    float *dst[kRefMaxCols];
    const float *append[kRefMaxCols];
    int rf[kRefMaxCols];
    int kind[kRefMaxCols];
    int n_cols, nb, n_samples, mask_append;
    int64_t N, n_keep, n_split, n_split_kept, n_dup_kept, n_append;
    const uint8_t *flags;
    const int32_t *book;
    const float *randn, *quats, *log_scales;
    const uint8_t *mask_in;
    uint8_t *mask_out;
};

// One split child's mean coordinate r: mean + R(q / |q|) (exp(log_scale) * z), as split_samples_kernel (csrc/densify.hip).
__device__ __forceinline__ float child_mean(const RefineMove &J, int64_t p, int64_t zrow, int r, float mean_r) {
    const float4 q = reinterpret_cast<const float4 *>(J.quats)[p];
    const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    const float w = q.x * inv, x = q.y * inv, y = q.z * inv, z = q.w * inv;
    const float v0 = expf(J.log_scales[p * 3]) * J.randn[zrow * 3], v1 = expf(J.log_scales[p * 3 + 1]) * J.randn[zrow * 3 + 1],
                v2 = expf(J.log_scales[p * 3 + 2]) * J.randn[zrow * 3 + 2];
    float r0, r1, r2;
    if (r == 0) { r0 = 1.f - 2.f * (y * y + z * z); r1 = 2.f * (x * y - w * z); r2 = 2.f * (x * z + w * y); }
    else if (r == 1) { r0 = 2.f * (x * y + w * z); r1 = 1.f - 2.f * (x * x + z * z); r2 = 2.f * (y * z - w * x); }
    else { r0 = 2.f * (x * z - w * y); r1 = 2.f * (y * z + w * x); r2 = 1.f - 2.f * (x * x + y * y); }
    return r0 * v0 + r1 * v1 + r2 * v2 + mean_r;
}

// One column's elements of this workgroup's 256 rows.  RF > 0: the row width as a constant (the division by it is the
// inner loop's most expensive operation otherwise); RF == 0: any width.
template <int RF>
__device__ __forceinline__ void move_column(const RefineMove &J, int c, int rf_any, int64_t row0, int rows, const int *s_old,
                                            const int *s_rank, const int *s_child, const int *s_dup, int64_t new0,
                                            int64_t dup0) {
    const int rf = RF > 0 ? RF : rf_any, kind = J.kind[c];
    const float *__restrict__ src = J.src[c] + row0 * rf;
    float *__restrict__ dst = J.dst[c];
    for (int i = threadIdx.x; i < rows * rf; i += 256) {
        const int r = i / rf, col = i - r * rf;
        const float v = src[i];
        const int o = s_old[r], ch = s_child[r], du = s_dup[r];
        if (o >= 0) dst[(int64_t)o * rf + col] = v;
        if (ch >= 0) {
            for (int s = 0; s < J.n_samples; ++s) {
                float nv = v;
                if (kind == COL_ZERO) nv = 0.f;
                else if (kind == COL_SCALES) nv = logf(shrink_scale(expf(v)));
                else if (kind == COL_MEANS) nv = child_mean(J, row0 + r, (int64_t)s * J.n_split + s_rank[r], col, v);
                dst[(new0 + (int64_t)s * J.n_split_kept + ch) * rf + col] = nv;
            }
        }
        if (du >= 0) {
            float nv = v;
            if (kind == COL_ZERO) nv = 0.f;
            else if (kind == COL_SCALES && s_rank[r] >= 0) nv = logf(shrink_scale(expf(v)));  // (split AND duplicated)
            dst[(dup0 + du) * rf + col] = nv;
        }
    }
}

// grid = (row blocks + append blocks, columns [+ 1 for the byte mask]): a workgroup moves ONE column of its 256 rows — a
// refinement of 75 000 Gaussians is 294 row blocks, which as 294 workgroups left most of the chip idle (measured: 100 us
// for 106 MB; the ranks from the flags are a few ballots, recomputing them per column costs nothing).
__global__ void __launch_bounds__(256) refine_move_kernel(RefineMove J) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = blockIdx.y;                                              // column; == n_cols: the byte mask
    const int64_t new0 = J.n_keep;                                         // first child row
    const int64_t dup0 = new0 + (int64_t)J.n_samples * J.n_split_kept;     // first duplicate row
    const int64_t app0 = dup0 + J.n_dup_kept;                              // first appended row
    if ((int)blockIdx.x >= J.nb) {  // appended rows: contiguous copies (or zeros)
        const int64_t j0 = (int64_t)(blockIdx.x - J.nb) * kRefRows;
        const int rows = (int)min((int64_t)kRefRows, J.n_append - j0);
        if (c < J.n_cols) {
            const int rf = J.rf[c];
            const bool zero = J.kind[c] == COL_ZERO;
            for (int i = tid; i < rows * rf; i += 256)
                J.dst[c][(app0 + j0) * rf + i] = zero ? 0.f : J.append[c][j0 * rf + i];
        } else if (tid < rows) {
            J.mask_out[app0 + j0 + tid] = (uint8_t)J.mask_append;
        }
        return;
    }
    __shared__ int wcnt[4][4];
    __shared__ int s_old[kRefRows], s_rank[kRefRows], s_child[kRefRows], s_dup[kRefRows];
    const int64_t row0 = (int64_t)blockIdx.x * kRefRows;
    const int rows = (int)min((int64_t)kRefRows, J.N - row0);
    const unsigned f = tid < rows ? J.flags[row0 + tid] : 0u;
    const unsigned want[4] = {RF_KEEP, RF_SPLIT, RF_SPLIT | RF_CHILD_KEEP, RF_DUP | RF_DUP_KEEP};
    const int seq[4] = {0, 1, 3, 4};  // (book's sequences: the DUP count is only a total)
    int rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long m = __ballot((f & want[k]) == want[k]);
        rank[k] = lanes_below(m);
        if (lane == 0) wcnt[w][k] = __popcll(m);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        for (int j = 0; j < w; ++j) rank[k] += wcnt[j][k];
        rank[k] += J.book[kRefBookHead + (int64_t)seq[k] * J.nb + blockIdx.x];
    }
    s_old[tid] = (f & RF_KEEP) ? rank[0] : -1;
    s_rank[tid] = (f & RF_SPLIT) ? rank[1] : -1;
    s_child[tid] = ((f & (RF_SPLIT | RF_CHILD_KEEP)) == (RF_SPLIT | RF_CHILD_KEEP)) ? rank[2] : -1;
    s_dup[tid] = ((f & (RF_DUP | RF_DUP_KEEP)) == (RF_DUP | RF_DUP_KEEP)) ? rank[3] : -1;
    __syncthreads();
    if (c >= J.n_cols) {
        if (tid < rows) {
            if (s_old[tid] >= 0) J.mask_out[s_old[tid]] = J.mask_in[row0 + tid];
            if (s_child[tid] >= 0)
                for (int s = 0; s < J.n_samples; ++s) J.mask_out[new0 + (int64_t)s * J.n_split_kept + s_child[tid]] = 0;
            if (s_dup[tid] >= 0) J.mask_out[dup0 + s_dup[tid]] = 0;
        }
        return;
    }
    switch (J.rf[c]) {
    case 1: move_column<1>(J, c, 1, row0, rows, s_old, s_rank, s_child, s_dup, new0, dup0); break;
    case 3: move_column<3>(J, c, 3, row0, rows, s_old, s_rank, s_child, s_dup, new0, dup0); break;
    case 4: move_column<4>(J, c, 4, row0, rows, s_old, s_rank, s_child, s_dup, new0, dup0); break;
    case 45: move_column<45>(J, c, 45, row0, rows, s_old, s_rank, s_child, s_dup, new0, dup0); break;
    default: move_column<0>(J, c, J.rf[c], row0, rows, s_old, s_rank, s_child, s_dup, new0, dup0); break;
    }
}

}  // namespace fsgs

using namespace fsgs;

extern "C" int64_t fsgs_refine_book_ints(int64_t N) {
    return kRefBookHead + (int64_t)kRefCounts * ((N + kRefRows - 1) / kRefRows);
}

extern "C" int fsgs_refine_mark(int64_t N, const fsgs_refine_rules *rules, const float *xys_grad_norm, const float *vis_counts,
                                const float *max_2Dsize, const float *means, const float *log_scales, const float *opac_logit,
                                const uint8_t *add_mask, const uint8_t *extra_cull, const uint64_t *hull_best,
                                const float *hull_points, const int32_t *hull_cells, const float *boxes, uint8_t *flags,
                                int32_t *book, int32_t *totals_mapped, fsgs_stream_t stream) {
    if (N < 0 || N > 0x7FFFFFFFll || !rules || !totals_mapped) return FSGS_EINVAL;
    if (rules->n_boxes < 0 || rules->n_samples < 0 || (rules->densify && rules->n_samples < 1)) return FSGS_EINVAL;
    if (N == 0) {  // nothing to launch: the totals are zero, and they have "landed"
        for (int k = 0; k < kRefCounts; ++k) totals_mapped[k] = 0;
        totals_mapped[7] = 1;
        return FSGS_OK;
    }
    if (!log_scales || !opac_logit || !flags || !book) return FSGS_EINVAL;
    if (rules->densify && (!xys_grad_norm || !vis_counts)) return FSGS_EINVAL;
    if ((hull_best || hull_points || rules->n_boxes > 0) && !means) return FSGS_EINVAL;
    if (hull_points && (hull_best || !hull_cells || rules->grid_dims[0] < 1 || rules->grid_dims[1] < 1 || rules->grid_dims[2] < 1 ||
                        (int64_t)rules->grid_dims[0] * rules->grid_dims[1] * rules->grid_dims[2] > (1ll << 24)))
        return FSGS_EINVAL;
    if (rules->n_boxes > 0 && !boxes) return FSGS_EINVAL;
    const int nb = ceil_div(N, kRefRows);
    if (hull_points)
        hipLaunchKernelGGL(refine_mark_kernel<4>, dim3(nb), dim3(1024), 0, as_stream(stream), N, nb, *rules, xys_grad_norm,
                           vis_counts, max_2Dsize, means, log_scales, opac_logit, add_mask, extra_cull,
                           reinterpret_cast<const unsigned long long *>(hull_best), reinterpret_cast<const float4 *>(hull_points),
                           hull_cells, boxes, flags, book, totals_mapped);
    else
        hipLaunchKernelGGL(refine_mark_kernel<1>, dim3(nb), dim3(256), 0, as_stream(stream), N, nb, *rules, xys_grad_norm,
                           vis_counts, max_2Dsize, means, log_scales, opac_logit, add_mask, extra_cull,
                           reinterpret_cast<const unsigned long long *>(hull_best), nullptr, hull_cells, boxes, flags, book,
                           totals_mapped);
    return check_launch();
}

extern "C" int fsgs_refine_move(int64_t N, int n_cols, const float *const *src, float *const *dst, const float *const *append,
                                const int *row_floats, const int *kind, const uint8_t *flags, const int32_t *book,
                                int64_t n_keep, int64_t n_split, int64_t n_split_kept, int64_t n_dup_kept, int64_t n_append,
                                int n_samples, const float *randn, const float *quats, const float *log_scales,
                                const uint8_t *mask_in, uint8_t *mask_out, int mask_append, fsgs_stream_t stream) {
    if (N < 0 || N > 0x7FFFFFFFll || n_cols < 0 || n_cols > kRefMaxCols) return FSGS_EINVAL;
    if (n_keep < 0 || n_keep > N || n_split < 0 || n_split_kept < 0 || n_split_kept > n_split || n_split > N || n_dup_kept < 0 ||
        n_dup_kept > N || n_append < 0 || n_samples < 0)
        return FSGS_EINVAL;
    const int64_t n_out = n_keep + (int64_t)n_samples * n_split_kept + n_dup_kept + n_append;
    if (n_out > 0x7FFFFFFFll) return FSGS_EINVAL;
    if (n_cols > 0 && (!src || !dst || !row_floats || !kind)) return FSGS_EINVAL;
    if (N > 0 && (!flags || !book)) return FSGS_EINVAL;
    if (mask_out && N > 0 && !mask_in) return FSGS_EINVAL;
    RefineMove J;
    J.n_cols = 0;
    bool any_means = false;
    for (int c = 0; c < n_cols; ++c) {
        if (row_floats[c] < 0 || kind[c] < COL_COPY || kind[c] > COL_SCALES) return FSGS_EINVAL;
        if (row_floats[c] == 0) continue;  // (features_rest is [N, 0, 3] at SH degree 0)
        if ((N > 0 && !src[c]) || (n_out > 0 && !dst[c])) return FSGS_EINVAL;
        if (n_append > 0 && kind[c] != COL_ZERO && (!append || !append[c])) return FSGS_EINVAL;
        if (kind[c] == COL_MEANS && row_floats[c] != 3) return FSGS_EINVAL;
        any_means = any_means || kind[c] == COL_MEANS;
        const int k = J.n_cols++;
        J.src[k] = src[c]; J.dst[k] = dst[c]; J.append[k] = append ? append[c] : nullptr;
        J.rf[k] = row_floats[c]; J.kind[k] = kind[c];
    }
    if (any_means && n_split_kept > 0 && (!randn || !quats || !log_scales)) return FSGS_EINVAL;
    J.nb = N > 0 ? ceil_div(N, kRefRows) : 0;
    J.n_samples = n_samples; J.mask_append = mask_append;
    J.N = N; J.n_keep = n_keep; J.n_split = n_split; J.n_split_kept = n_split_kept; J.n_dup_kept = n_dup_kept;
    J.n_append = n_append;
    J.flags = flags; J.book = book; J.randn = randn; J.quats = quats; J.log_scales = log_scales;
    J.mask_in = mask_in; J.mask_out = mask_out;
    const int blocks = J.nb + ceil_div(n_append, kRefRows);
    const int columns = J.n_cols + (mask_out ? 1 : 0);
    if (blocks == 0 || columns == 0) return FSGS_OK;
    hipLaunchKernelGGL(refine_move_kernel, dim3(blocks, columns), dim3(256), 0, as_stream(stream), J);
    return check_launch();
}
