// N2: the photometric loss on the render as two kernels instead of ~40 torch
// launches (5 depthwise convolutions forward, their transposes backward).
//   loss = (1-l) * mean|pred-gt| + l * (1 - SSIM11(pred, gt))
// as FusionSense computes it (/root/reference/dn_splatter/dn_model.py:683 -> splatfacto main loss
// with the torchmetrics StructuralSimilarityIndexMeasure(kernel_size=11) swapped in at :244):
// Gaussian 11x11 window, sigma 1.5, the SSIM map averaged over the (H-10)x(W-10) interior.
//
// One workgroup (256 threads) per 32x32 output tile and channel: the 42x42 halo of pred and gt is staged
// in LDS once and blurred separably (rows, then columns) for the five moments; every thread produces FOUR
// adjacent outputs per pass from a sliding window of 14 inputs held in registers (14 LDS reads per map
// instead of 44: with one output per thread the kernel was bound by its ~90 LDS reads per pixel), then the
// SSIM value plus its three partial derivatives (w.r.t. mu1, sigma1^2, sigma12) per pixel.  The backward
// blurs those three maps with the same window:
//   dSSIM/dpred = G*(dm_dmu1) + 2 pred G*(dm_dsigma1sq) + gt G*(dm_dsigma12).
// Layout: images are [H,W,3] (channel-last, as the renderer writes them); the three saved
// derivative maps are planar [3,H,W] so that their writes and the backward's halo reads coalesce.
#include "common.h"

namespace fsgs {

constexpr int kLT = 32;             // output tile edge
constexpr int kLR = 5;              // window radius
constexpr int kLH = kLT + 2 * kLR;  // 42
constexpr int kLP = kLH + 1;        // halo row pitch (odd: the row pass reads 8 rows x 8 column groups per wave)
constexpr int kLQ = kLT + 1;        // blurred-row pitch (odd: conflict-free for the row pass's writes and the column pass's reads)
constexpr int kHaloIters = (kLH * kLH + 255) / 256;

__constant__ float kGauss11[11] = {0.0010283801f, 0.0075987581f, 0.0360007721f, 0.1093606895f,
                                   0.2130055377f, 0.2660117249f, 0.2130055377f, 0.1093606895f,
                                   0.0360007721f, 0.0075987581f, 0.0010283801f};

__device__ __forceinline__ float block_sum_256(float v, float *lds4) {
    v = wave_sum_to_last_row(v);
    const int tr = threadIdx.x;
    if ((tr & 63) == 63) lds4[tr >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// Which (tile, channel) a workgroup of the 1-D grid takes.  Block b runs on XCD b % 8 (observed placement: speed only), each
// XCD with its own L2: the three channel workgroups of a tile read the same cache lines of the channel-last images and
// neighbouring tiles share 10 halo columns / rows, so every XCD gets a CONTIGUOUS run of (tile, channel) ids, channels
// fastest, tiles row-major — dispatched back to back on that XCD.  (Round 5: with channels in blockIdx.z the three
// readers of a line ran a third of the launch apart, on different XCDs: FETCH 75 MB forward / 120 MB backward at 800x800
// for 15 / 38 MB of distinct input; 24.6 -> 13.1 us backward, 22.9 -> 20.9 us forward.)  ``extra`` ids beyond the tiles
// are returned with ch = 3 (the backward's combine).
__device__ __forceinline__ bool loss_tile_of_block(int gx, int gy, int extra, int &ch, int &tx, int &ty) {
    const int tiles3 = gx * gy * 3, total = tiles3 + extra;
    const int per = (total + 7) >> 3;
    const int id = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (id >= total) return false;
    if (id >= tiles3) { ch = 3; tx = ty = 0; return true; }
    ch = id % 3;
    const int t = id / 3;
    ty = t / gx; tx = t - ty * gx;
    return true;
}
__host__ inline unsigned loss_grid_blocks(int H, int W, int extra) {
    const int total = (int)(ceil_div(W, kLT) * ceil_div(H, kLT)) * 3 + extra;
    return 8u * (unsigned)((total + 7) >> 3);
}

// One pixel's SSIM value and its derivatives by mu1, sigma1^2, sigma12 from the five blurred moments (mu1, mu2, E[pp], E[gg],
// E[pg]).  No contraction: what the optimiser fuses depends on the code around the call, and the two instantiations of the
// forward kernel (and the short way of its all-masked tiles) must give the same bits on the same moments.
__device__ __forceinline__ void ssim_point(const float (&mom)[5], float &ssim, float &d_mu1, float &d_s1, float &d_s12) {
#pragma clang fp contract(off)
    const float mu1 = mom[0], mu2 = mom[1], e11 = mom[2], e22 = mom[3], e12 = mom[4];
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float A = mu1_sq + mu2_sq + C1, B = s1 + s2 + C2;
    const float C = 2.f * mu12 + C1, D = 2.f * s12 + C2;
    // (two reciprocals instead of eight IEEE divisions: 1 ulp each, far inside the parity tolerance)
    const float rA = __builtin_amdgcn_rcpf(A), rB = __builtin_amdgcn_rcpf(B);
    const float X = rA * rB, CDX = C * D * X;
    ssim = CDX;
    d_mu1 = 2.f * (mu2 * (D - C) * X + mu1 * CDX * (rB - rA));
    d_s1 = -CDX * rB;
    d_s12 = 2.f * C * X;
}

// partials[2*blk] = sum |pred-gt| of this (tile, channel), partials[2*blk+1] = sum of its interior SSIM values
template <bool MASKED>
__global__ void __launch_bounds__(256)
ssim_l1_fwd_kernel(int H, int W, const float *__restrict__ pred, const float *__restrict__ gt,
                   const float *__restrict__ mask, float *__restrict__ dm_dmu1, float *__restrict__ dm_dsigma1,
                   float *__restrict__ dm_dsigma12, float *__restrict__ sums) {
    // ONE arena: the two halos first, the five row-blurred maps over them once every thread holds its row results in
    // registers (27.7 KB instead of 48: five workgroups per CU instead of three — the kernel is bound by its three
    // dependent phases, not by bytes or instructions; round 4)
    constexpr int kArena = (5 * kLH * kLQ > 2 * kLH * kLP) ? 5 * kLH * kLQ : 2 * kLH * kLP;
    __shared__ float arena[kArena];
    __shared__ float red[4];
    float (*sp)[kLP] = reinterpret_cast<float (*)[kLP]>(arena);
    float (*sg)[kLP] = reinterpret_cast<float (*)[kLP]>(arena + kLH * kLP);
    auto hb = [&](int m, int ly, int lx) -> float & { return arena[(m * kLH + ly) * kLQ + lx]; };
    const int gx = (W + kLT - 1) / kLT, gy = (H + kLT - 1) / kLT;
    int ch, tx, ty;
    if (!loss_tile_of_block(gx, gy, 0, ch, tx, ty)) return;
    const int x0 = tx * kLT, y0 = ty * kLT;
    const int tr = threadIdx.x;
    // mask (FusionSense object mask, nullable): both images are multiplied by it first (splatfacto get_loss_dict, called
    // at /root/reference/dn_splatter/dn_model.py:683).  An object mask leaves most of a frame at zero: a tile whose
    // whole halo is masked out has all-zero inputs — it skips the image loads and both blur passes and runs the
    // per-pixel tail on zero moments (the same instructions on the same values: the same bits as the long way).
    float hm[MASKED ? kHaloIters : 1];
    bool live = true;
    if (MASKED) {
        bool any = false;
#pragma unroll
        for (int it = 0; it < kHaloIters; ++it) {
            const int i = tr + it * 256;
            const int ly = i / kLH, lx = i - ly * kLH;
            const int y = y0 + ly - kLR, x = x0 + lx - kLR;
            hm[it] = 0.f;
            if (i < kLH * kLH && y >= 0 && y < H && x >= 0 && x < W) hm[it] = mask[(int64_t)y * W + x];
            any |= hm[it] != 0.f;
        }
        live = __syncthreads_or(any);
    }
    const int lx = tr & 31, ry = (tr >> 5) * 4;
    float own_d[4], mom[4][5];
    if (live) {
    // the halo: every load of this thread is issued before the first LDS write (written as one loop the
    // compiler waits for each element in turn: serialised HBM round trips)
    float hp[kHaloIters], hg[kHaloIters];
#pragma unroll
    for (int it = 0; it < kHaloIters; ++it) {
        const int i = tr + it * 256;
        const int ly = i / kLH, lx = i - ly * kLH;
        const int y = y0 + ly - kLR, x = x0 + lx - kLR;
        hp[it] = 0.f; hg[it] = 0.f;
        if (i < kLH * kLH && y >= 0 && y < H && x >= 0 && x < W) {
            if (MASKED) {
                const float m = hm[it];
                hp[it] = pred[((int64_t)y * W + x) * 3 + ch] * m;
                hg[it] = gt[((int64_t)y * W + x) * 3 + ch] * m;
            } else {
                hp[it] = pred[((int64_t)y * W + x) * 3 + ch];
                hg[it] = gt[((int64_t)y * W + x) * 3 + ch];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < kHaloIters; ++it) {
        const int i = tr + it * 256;
        const int ly = i / kLH, lx = i - ly * kLH;
        if (i < kLH * kLH) {
            sp[ly][lx] = hp[it];
            sg[ly][lx] = hg[it];
        }
    }
    __syncthreads();
    // row pass: 42 rows x 8 groups of 4 columns (336 items, at most two per thread); the 14 inputs of a group slide
    // through registers, the results stay in registers until every thread has read its inputs
    constexpr int kRowItems = kLH * (kLT / 4), kRowReps = (kRowItems + 255) / 256;
    float acc[kRowReps][4][5];
#pragma unroll
    for (int rep = 0; rep < kRowReps; ++rep) {
        const int item = tr + rep * 256;
        const int ly = min(item, kRowItems - 1) >> 3, cx = (item & 7) * 4;
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int m = 0; m < 5; ++m) acc[rep][o][m] = 0.f;
        if (item < kRowItems)
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            const float p = sp[ly][cx + t], g = sg[ly][cx + t];
            const float pp = p * p, gg = g * g, pg = p * g;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int k = t - o;
                if (k >= 0 && k < 11) {
                    const float w = kGauss11[k];
                    acc[rep][o][0] += w * p; acc[rep][o][1] += w * g; acc[rep][o][2] += w * pp;
                    acc[rep][o][3] += w * gg; acc[rep][o][4] += w * pg;
                }
            }
        }
    }
    // column pass: thread = (column lx, group of 4 rows); its own four pixels for the L1 term
#pragma unroll
    for (int o = 0; o < 4; ++o) own_d[o] = fabsf(sp[ry + o + kLR][lx + kLR] - sg[ry + o + kLR][lx + kLR]);
    __syncthreads();  // every input read: the arena now takes the row-blurred maps
#pragma unroll
    for (int rep = 0; rep < kRowReps; ++rep) {
        const int item = tr + rep * 256;
        if (item < kRowItems) {
            const int ly = item >> 3, cx = (item & 7) * 4;
#pragma unroll
            for (int m = 0; m < 5; ++m)
#pragma unroll
                for (int o = 0; o < 4; ++o) hb(m, ly, cx + o) = acc[rep][o][m];
        }
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int m = 0; m < 5; ++m) mom[o][m] = 0.f;
#pragma unroll
    for (int t = 0; t < 14; ++t) {
        float v[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) v[m] = hb(m, ry + t, lx);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int k = t - o;
            if (k >= 0 && k < 11) {
                const float w = kGauss11[k];
#pragma unroll
                for (int m = 0; m < 5; ++m) mom[o][m] += w * v[m];
            }
        }
    }
    } else {
        // (a zero the optimiser cannot see through: the tail below must stay the code that runs on the device for
        // every tile, not a constant folded at compile time with another rounding of the reciprocals)
        float z = 0.f;
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            own_d[o] = z;
#pragma unroll
            for (int m = 0; m < 5; ++m) mom[o][m] = z;
        }
    }
    const int x = x0 + lx;
    float l1 = 0.f, ssim = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int ly = ry + o, y = y0 + ly;
        const bool in_img = (x < W) && (y < H);
        const bool interior = in_img && x >= kLR && x < W - kLR && y >= kLR && y < H - kLR;
        float d_mu1 = 0.f, d_s1 = 0.f, d_s12 = 0.f;
        if (in_img) l1 += own_d[o];
        if (interior) {
            float v;
            ssim_point(mom[o], v, d_mu1, d_s1, d_s12);
            ssim += v;
        }
        if (in_img) {
            const int64_t oo = ((int64_t)ch * H + y) * W + x;  // planar [3,H,W]: coalesced rows
            dm_dmu1[oo] = d_mu1;
            dm_dsigma1[oo] = d_s1;
            dm_dsigma12[oo] = d_s12;
        }
    }
    const float t_l1 = block_sum_256(l1, red);
    __syncthreads();
    const float t_ss = block_sum_256(ssim, red);
    if (tr == 255) {  // per-workgroup partials: same-address float atomics from thousands of blocks serialise
        const int blk = (ch * gy + ty) * gx + tx;
        sums[2 * blk + 0] = t_l1;
        sums[2 * blk + 1] = t_ss;
    }
}

// out[0] = bias + sum_t (w[2t] * colsum0(partials_t) + w[2t+1] * colsum1(partials_t)): the scalar loss
// from the per-workgroup partials of up to four fused terms, in one launch and without a host sync.
constexpr int kCombineMaxTerms = 6;
constexpr int kCombineMaxCols = 8;
struct CombineArgs {
    const float *partials[kCombineMaxTerms];  // [rows, cols] each
    long long rows[kCombineMaxTerms];
    int cols[kCombineMaxTerms];
    float w[kCombineMaxTerms][kCombineMaxCols];  // weight of every column
    int n_terms;
    float bias;
};

// the combine on THREADS threads of one workgroup (f64 accumulation)
template <int THREADS>
__device__ __forceinline__ void combine_partials(const CombineArgs &a, float *__restrict__ out, double *red) {
    double acc = 0.0;
    for (int t = 0; t < a.n_terms; ++t) {
        const float *p = a.partials[t];
        const int cols = a.cols[t];
        for (long long r = threadIdx.x; r < a.rows[t]; r += THREADS)
            for (int c = 0; c < cols; ++c) acc += (double)p[r * cols + c] * (double)a.w[t][c];
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < THREADS / 64; ++k) tot += red[k];
        out[0] = (float)(tot + (double)a.bias);
    }
}

// v_pred = g_l1 * sign(pred-gt) + g_ssim * (G*dm_dmu1 + 2 pred G*dm_dsigma1 + gt G*dm_dsigma12)
template <bool MASKED>
__global__ void __launch_bounds__(256)
ssim_l1_bwd_kernel(int H, int W, const float *__restrict__ pred, const float *__restrict__ gt,
                   const float *__restrict__ mask, const float *__restrict__ dm_dmu1, const float *__restrict__ dm_dsigma1,
                   const float *__restrict__ dm_dsigma12, const float *__restrict__ v_loss, float g_l1,
                   float g_ssim, float *__restrict__ v_pred, CombineArgs comb, float *__restrict__ loss_out) {
    // (one arena, as in the forward: the three halos, then the three row-blurred maps over them — 21.7 KB instead of 41.8)
    constexpr int kArena = (3 * kLH * kLQ > 3 * kLH * kLP) ? 3 * kLH * kLQ : 3 * kLH * kLP;
    __shared__ float arena[kArena];
    __shared__ double comb_red[4];
    auto sm = [&](int m, int ly, int lx) -> float & { return arena[(m * kLH + ly) * kLP + lx]; };
    auto hb = [&](int m, int ly, int lx) -> float & { return arena[(m * kLH + ly) * kLQ + lx]; };
    const int gx = (W + kLT - 1) / kLT, gy = (H + kLT - 1) / kLT;
    int ch, tx, ty;
    if (!loss_tile_of_block(gx, gy, loss_out ? 1 : 0, ch, tx, ty)) return;
    if (ch == 3) {
        // one workgroup more (when asked for): it combines the loss partials into the scalar loss — the value is only
        // reported, so it rides in this launch instead of one of its own
        combine_partials<256>(comb, loss_out, comb_red);
        return;
    }
    const int x0 = tx * kLT, y0 = ty * kLT;
    const int tr = threadIdx.x;
    // this thread's own four pixels
    const int lx = tr & 31, ry = (tr >> 5) * 4;
    const int x = x0 + lx;
    float own_m[4];
    if (MASKED) {
        // a tile whose own pixels are all masked out has v_pred = mask * (...) = 0: written without reading anything else
        bool any = false;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int y = y0 + ry + o;
            own_m[o] = mask[(int64_t)min(y, H - 1) * W + min(x, W - 1)];
            any |= own_m[o] != 0.f;
        }
        if (!__syncthreads_or(any)) {
            if (x < W)
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int y = y0 + ry + o;
                    if (y < H) v_pred[((int64_t)y * W + x) * 3 + ch] = 0.f;
                }
            return;
        }
    }
    float ha[kHaloIters], hbv[kHaloIters], hc[kHaloIters];  // (all loads first, as in the forward)
#pragma unroll
    for (int it = 0; it < kHaloIters; ++it) {
        const int i = tr + it * 256;
        const int ly = i / kLH, lx = i - ly * kLH;
        const int y = y0 + ly - kLR, x = x0 + lx - kLR;
        ha[it] = 0.f; hbv[it] = 0.f; hc[it] = 0.f;
        if (i < kLH * kLH && y >= 0 && y < H && x >= 0 && x < W) {
            const int64_t o = ((int64_t)ch * H + y) * W + x;
            ha[it] = dm_dmu1[o]; hbv[it] = dm_dsigma1[o]; hc[it] = dm_dsigma12[o];
        }
    }
    // the images at its own pixels and the upstream scalar, in flight with the halo
    float own_p[4], own_g[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int y = y0 + ry + o;
        const int64_t op = (int64_t)min(y, H - 1) * W + min(x, W - 1);
        if (!MASKED) own_m[o] = 1.f;
        own_p[o] = MASKED ? pred[op * 3 + ch] * own_m[o] : pred[op * 3 + ch];
        own_g[o] = MASKED ? gt[op * 3 + ch] * own_m[o] : gt[op * 3 + ch];
    }
    const float up = v_loss[0];
#pragma unroll
    for (int it = 0; it < kHaloIters; ++it) {
        const int i = tr + it * 256;
        const int ly = i / kLH, lxx = i - ly * kLH;
        if (i < kLH * kLH) {
            sm(0, ly, lxx) = ha[it]; sm(1, ly, lxx) = hbv[it]; sm(2, ly, lxx) = hc[it];
        }
    }
    __syncthreads();
    constexpr int kRowItems = kLH * (kLT / 4), kRowReps = (kRowItems + 255) / 256;
    float acc[kRowReps][4][3];
#pragma unroll
    for (int rep = 0; rep < kRowReps; ++rep) {
        const int item = tr + rep * 256;
        const int ly = min(item, kRowItems - 1) >> 3, cx = (item & 7) * 4;
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[rep][o][m] = 0.f;
        if (item < kRowItems)
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            float v[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) v[m] = sm(m, ly, cx + t);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int k = t - o;
                if (k >= 0 && k < 11) {
                    const float w = kGauss11[k];
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc[rep][o][m] += w * v[m];
                }
            }
        }
    }
    __syncthreads();  // every input read: the arena now takes the row-blurred maps
#pragma unroll
    for (int rep = 0; rep < kRowReps; ++rep) {
        const int item = tr + rep * 256;
        if (item < kRowItems) {
            const int ly = item >> 3, cx = (item & 7) * 4;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int o = 0; o < 4; ++o) hb(m, ly, cx + o) = acc[rep][o][m];
        }
    }
    __syncthreads();
    float out[4][3];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int m = 0; m < 3; ++m) out[o][m] = 0.f;
#pragma unroll
    for (int t = 0; t < 14; ++t) {
        float v[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) v[m] = hb(m, ry + t, lx);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int k = t - o;
            if (k >= 0 && k < 11) {
                const float w = kGauss11[k];
#pragma unroll
                for (int m = 0; m < 3; ++m) out[o][m] += w * v[m];
            }
        }
    }
    if (x >= W) return;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int y = y0 + ry + o;
        if (y >= H) break;
        const float p = own_p[o], g = own_g[o];
        const float d = p - g;
        const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
        const float gv = up * (g_l1 * sgn + g_ssim * (out[o][0] + 2.f * p * out[o][1] + g * out[o][2]));
        v_pred[((int64_t)y * W + x) * 3 + ch] = MASKED ? own_m[o] * gv : gv;
    }
}

// Auxiliary L1 terms of the FusionSense loss on the depth and normal images
// (/root/reference/dn_splatter/dn_model.py:673-925: sensor-depth and normal supervision), one pass:
// partial[blk] = (sum |depth - depth_gt|, sum |normal - normal_gt|); the backward writes
// w * sign(.) / count straight into the two gradient images.
// GRADS: the gradient images are written in the same pass (the caller knows the upstream gradient of the
// loss already — the trainer's tape-free step seeds it itself): one launch and one read of the images less.
template <bool GRADS>
__global__ void __launch_bounds__(256)
aux_l1_fwd_kernel(int64_t P, const float *__restrict__ depth, const float *__restrict__ depth_gt,
                  const float *__restrict__ normal, const float *__restrict__ normal_gt,
                  float *__restrict__ partial, const float *__restrict__ v_loss, float g_depth, float g_normal,
                  float *__restrict__ v_depth, float *__restrict__ v_normal) {
    __shared__ float red[4];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float sd = 0.f, sn = 0.f;
    if (p < P) {
        const float up = GRADS ? v_loss[0] : 0.f;
        const float d = depth[p] - depth_gt[p];
        sd = fabsf(d);
        if (GRADS) v_depth[p] = up * g_depth * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
        if (normal) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float e = normal[p * 3 + k] - normal_gt[p * 3 + k];
                sn += fabsf(e);
                if (GRADS) v_normal[p * 3 + k] = up * g_normal * ((e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f));
            }
        }
    }
    const float td = block_sum_256(sd, red);
    __syncthreads();
    const float tn = block_sum_256(sn, red);
    if (threadIdx.x == 255) {
        partial[2 * blockIdx.x + 0] = td;
        partial[2 * blockIdx.x + 1] = tn;
    }
}

__global__ void __launch_bounds__(256)
aux_l1_bwd_kernel(int64_t P, const float *__restrict__ depth, const float *__restrict__ depth_gt,
                  const float *__restrict__ normal, const float *__restrict__ normal_gt,
                  const float *__restrict__ v_loss, float g_depth, float g_normal,
                  float *__restrict__ v_depth, float *__restrict__ v_normal) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float up = v_loss[0];
    const float d = depth[p] - depth_gt[p];
    v_depth[p] = up * g_depth * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
    if (normal) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float e = normal[p * 3 + k] - normal_gt[p * 3 + k];
            v_normal[p * 3 + k] = up * g_normal * ((e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f));
        }
    }
}

__global__ void __launch_bounds__(1024) loss_combine_kernel(CombineArgs a, float *__restrict__ out) {
    __shared__ double red[16];
    combine_partials<1024>(a, out, red);
}

}  // namespace fsgs

using namespace fsgs;

static int fill_combine_args(CombineArgs &a, int n_terms, const float *const *partials, const int64_t *rows,
                             const int *cols, const float *weights, float bias) {
    // cols == NULL: two columns per term and two weights per term (the original layout)
    if (n_terms < 0 || n_terms > kCombineMaxTerms) return FSGS_EINVAL;
    if (n_terms && (!partials || !rows || !weights)) return FSGS_EINVAL;
    const float *w = weights;
    for (int t = 0; t < kCombineMaxTerms; ++t) {
        const bool on = t < n_terms;
        const int c = on ? (cols ? cols[t] : 2) : 0;
        if (on && (rows[t] < 0 || c < 1 || c > kCombineMaxCols || (rows[t] > 0 && !partials[t]))) return FSGS_EINVAL;
        a.partials[t] = on ? partials[t] : nullptr;
        a.rows[t] = on ? rows[t] : 0;
        a.cols[t] = c;
        for (int k = 0; k < kCombineMaxCols; ++k) a.w[t][k] = (k < c) ? *w++ : 0.f;
    }
    a.n_terms = n_terms;
    a.bias = bias;
    return FSGS_OK;
}

extern "C" int fsgs_loss_combine(int n_terms, const float *const *partials, const int64_t *rows,
                                 const float *weights, float bias, float *out, fsgs_stream_t stream) {
    return fsgs_loss_combine_cols(n_terms, partials, rows, nullptr, weights, bias, out, stream);
}

extern "C" int fsgs_loss_combine_cols(int n_terms, const float *const *partials, const int64_t *rows, const int *cols,
                                      const float *weights, float bias, float *out, fsgs_stream_t stream) {
    if (!out) return FSGS_EINVAL;
    CombineArgs a;
    const int rc = fill_combine_args(a, n_terms, partials, rows, cols, weights, bias);
    if (rc != FSGS_OK) return rc;
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(1024), 0, as_stream(stream), a, out);
    return check_launch();
}

extern "C" int fsgs_aux_l1_fwd(int64_t n_pixels, const float *depth, const float *depth_gt, const float *normal,
                               const float *normal_gt, float *partial, fsgs_stream_t stream) {
    if (n_pixels < 0) return FSGS_EINVAL;
    if (n_pixels == 0) return FSGS_OK;
    if (!depth || !depth_gt || !partial || (normal && !normal_gt)) return FSGS_EINVAL;
    hipLaunchKernelGGL(aux_l1_fwd_kernel<false>, dim3(ceil_div(n_pixels, 256)), dim3(256), 0, as_stream(stream),
                       n_pixels, depth, depth_gt, normal, normal_gt, partial, nullptr, 0.f, 0.f, nullptr, nullptr);
    return check_launch();
}

extern "C" int fsgs_aux_l1_fwd_bwd(int64_t n_pixels, const float *depth, const float *depth_gt, const float *normal,
                                   const float *normal_gt, float *partial, const float *v_loss, float g_depth,
                                   float g_normal, float *v_depth, float *v_normal, fsgs_stream_t stream) {
    if (n_pixels < 0) return FSGS_EINVAL;
    if (n_pixels == 0) return FSGS_OK;
    if (!depth || !depth_gt || !partial || !v_loss || !v_depth || (normal && (!normal_gt || !v_normal)))
        return FSGS_EINVAL;
    hipLaunchKernelGGL(aux_l1_fwd_kernel<true>, dim3(ceil_div(n_pixels, 256)), dim3(256), 0, as_stream(stream),
                       n_pixels, depth, depth_gt, normal, normal_gt, partial, v_loss, g_depth, g_normal, v_depth,
                       v_normal);
    return check_launch();
}

extern "C" int fsgs_aux_l1_bwd(int64_t n_pixels, const float *depth, const float *depth_gt, const float *normal,
                               const float *normal_gt, const float *v_loss, float g_depth, float g_normal,
                               float *v_depth, float *v_normal, fsgs_stream_t stream) {
    if (n_pixels < 0) return FSGS_EINVAL;
    if (n_pixels == 0) return FSGS_OK;
    if (!depth || !depth_gt || !v_loss || !v_depth || (normal && (!normal_gt || !v_normal))) return FSGS_EINVAL;
    hipLaunchKernelGGL(aux_l1_bwd_kernel, dim3(ceil_div(n_pixels, 256)), dim3(256), 0, as_stream(stream), n_pixels,
                       depth, depth_gt, normal, normal_gt, v_loss, g_depth, g_normal, v_depth, v_normal);
    return check_launch();
}

extern "C" int64_t fsgs_ssim_l1_num_partials(int H, int W) {
    return (int64_t)ceil_div(W, kLT) * ceil_div(H, kLT) * 3;
}

extern "C" int fsgs_ssim_l1_fwd(int H, int W, const float *pred, const float *gt, float *dm_dmu1,
                                float *dm_dsigma1, float *dm_dsigma12, float *sums, fsgs_stream_t stream) {
    return fsgs_ssim_l1_fwd_masked(H, W, pred, gt, nullptr, dm_dmu1, dm_dsigma1, dm_dsigma12, sums, stream);
}

extern "C" int fsgs_ssim_l1_fwd_masked(int H, int W, const float *pred, const float *gt, const float *mask,
                                       float *dm_dmu1, float *dm_dsigma1, float *dm_dsigma12, float *sums,
                                       fsgs_stream_t stream) {
    if (H < 11 || W < 11) return FSGS_EINVAL;
    if (!pred || !gt || !dm_dmu1 || !dm_dsigma1 || !dm_dsigma12 || !sums) return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
    if (mask)
        hipLaunchKernelGGL(ssim_l1_fwd_kernel<true>, dim3(loss_grid_blocks(H, W, 0)), dim3(256), 0, s,
                           H, W, pred, gt, mask, dm_dmu1, dm_dsigma1, dm_dsigma12, sums);
    else
        hipLaunchKernelGGL(ssim_l1_fwd_kernel<false>, dim3(loss_grid_blocks(H, W, 0)), dim3(256), 0, s,
                           H, W, pred, gt, mask, dm_dmu1, dm_dsigma1, dm_dsigma12, sums);
    return check_launch();
}

extern "C" int fsgs_ssim_l1_bwd(int H, int W, const float *pred, const float *gt, const float *dm_dmu1,
                                const float *dm_dsigma1, const float *dm_dsigma12, const float *v_loss,
                                float g_l1, float g_ssim, float *v_pred, fsgs_stream_t stream) {
    return fsgs_ssim_l1_bwd_masked(H, W, pred, gt, nullptr, dm_dmu1, dm_dsigma1, dm_dsigma12, v_loss, g_l1, g_ssim,
                                   v_pred, 0, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, stream);
}

// fsgs_ssim_l1_bwd + fsgs_loss_combine in one launch (the scalar loss is only reported: one workgroup of an extra
// grid slice sums the partials while the others run the backward)
extern "C" int fsgs_ssim_l1_bwd_combine(int H, int W, const float *pred, const float *gt, const float *dm_dmu1,
                                        const float *dm_dsigma1, const float *dm_dsigma12, const float *v_loss,
                                        float g_l1, float g_ssim, float *v_pred, int n_terms,
                                        const float *const *partials, const int64_t *rows, const float *weights,
                                        float bias, float *loss_out, fsgs_stream_t stream) {
    if (!loss_out) return FSGS_EINVAL;
    return fsgs_ssim_l1_bwd_masked(H, W, pred, gt, nullptr, dm_dmu1, dm_dsigma1, dm_dsigma12, v_loss, g_l1, g_ssim,
                                   v_pred, n_terms, partials, rows, nullptr, weights, bias, loss_out, stream);
}

// The general form: optional pixel mask (v_pred = mask * d loss / d(pred * mask)), optional combine slice
// (loss_out != NULL) over up to 6 partial-sum tables of up to 8 weighted columns each.
extern "C" int fsgs_ssim_l1_bwd_masked(int H, int W, const float *pred, const float *gt, const float *mask,
                                       const float *dm_dmu1, const float *dm_dsigma1, const float *dm_dsigma12,
                                       const float *v_loss, float g_l1, float g_ssim, float *v_pred, int n_terms,
                                       const float *const *partials, const int64_t *rows, const int *cols,
                                       const float *weights, float bias, float *loss_out, fsgs_stream_t stream) {
    if (H < 11 || W < 11) return FSGS_EINVAL;
    if (!pred || !gt || !dm_dmu1 || !dm_dsigma1 || !dm_dsigma12 || !v_loss || !v_pred) return FSGS_EINVAL;
    CombineArgs a{};
    if (loss_out) {
        const int rc = fill_combine_args(a, n_terms, partials, rows, cols, weights, bias);
        if (rc != FSGS_OK) return rc;
    }
    const dim3 grid(loss_grid_blocks(H, W, loss_out ? 1 : 0));
    if (mask)
        hipLaunchKernelGGL(ssim_l1_bwd_kernel<true>, grid, dim3(256), 0, as_stream(stream), H, W, pred, gt, mask,
                           dm_dmu1, dm_dsigma1, dm_dsigma12, v_loss, g_l1, g_ssim, v_pred, a, loss_out);
    else
        hipLaunchKernelGGL(ssim_l1_bwd_kernel<false>, grid, dim3(256), 0, as_stream(stream), H, W, pred, gt, mask,
                           dm_dmu1, dm_dsigma1, dm_dsigma12, v_loss, g_l1, g_ssim, v_pred, a, loss_out);
    return check_launch();
}
