// One training step as one call (include/fsgs.h: fsgs_step_plan).  No kernel lives here: the three entry points enqueue
// the per-op entry points of this library back to back from one caller-owned argument block, in the order and with the
// arguments fusionsense_amd/fused.py's tape-free step passes them (get_outputs of /root/reference/dn_splatter/dn_model.py:
// 469-671 on the count-free binning route, the benchmark loss, both backward passes, after_train statistics and Adam).
// What it removes is host time: ~0.45 ms of interpreter + ctypes work per step against ~8 launches x a few microseconds.
#include <chrono>

#include "common.h"

namespace {

inline int bad_plan(const fsgs_step_plan *p) {
    if (!p) return 1;
    if (p->n <= 0 || p->width <= 0 || p->height <= 0 || p->tile_width <= 0 || p->tile_height <= 0 || p->capacity <= 0) return 1;
    if (p->sh_degree < 0 || p->sh_degree > 3) return 1;
    if (!p->means || !p->quats || !p->log_scales || !p->opac_logit || !p->features_dc || !p->features_rest) return 1;
    if (!p->viewmat || !p->K || !p->campos || !p->c2w || !p->background) return 1;
    if (!p->scales_exp || !p->opac_sig || !p->radii || !p->means2d || !p->depths || !p->conics) return 1;
    if (!p->tiles_per_gauss || !p->isect_offsets || !p->bucket_base || !p->tile_cursor || !p->buckets || !p->next_bucket_base) return 1;
    if (!p->mapped || !p->packed || !p->normals_world || !p->zero_cells || !p->payload) return 1;
    if (!p->render || !p->alphas || !p->last_ids || !p->render_extra || !p->records || !p->n_rec || !p->seg_state || !p->seg_split) return 1;
    if (!p->rgb || !p->depth || !p->normal || !p->gt_rgb || !p->seed || !p->v_depth_img) return 1;
    if (p->loss_kind == 0) {
        if (!p->gt_depth || !p->aux_partial || (p->gt_normal && !p->v_normal_img)) return 1;
    } else if (p->loss_kind == 1) {
        if (!p->sensor_depth || !p->fa_partial || !p->v_normal_img) return 1;
        if (p->n_touch < 0 || (p->n_touch > 0 && (!p->touch_idx || !p->touch_normals || !p->touch_partial))) return 1;
    } else {
        return 1;
    }
    if (!p->ssim_maps || !p->ssim_sums || !p->v_rgb || !p->loss_out || !p->v_packed || !p->absgrad) return 1;
    if (p->adam.n_groups != 6 && p->adam.n_groups != 0) return 1;
    if (p->adam.n_groups == 0 && (!p->g_means || !p->g_log_scales || !p->g_quats || !p->g_features_dc || !p->g_features_rest ||
                                  !p->g_opac_logit))
        return 1;
    if (p->bwd_order && (!p->bwd_queue || !p->order_counters)) return 1;
    return 0;
}

// the caller's timing hooks around launch k
struct Hook {
    const fsgs_step_plan *p;
    hipStream_t s;
    int k;
    Hook(const fsgs_step_plan *p_, hipStream_t s_, int k_) : p(p_), s(s_), k(k_) {
        if (p->ev_before[k]) (void)hipEventRecord((hipEvent_t)p->ev_before[k], s);
    }
    ~Hook() {
        if (p->ev_after[k]) (void)hipEventRecord((hipEvent_t)p->ev_after[k], s);
    }
};

}  // namespace

extern "C" int64_t fsgs_step_plan_bytes(void) { return (int64_t)sizeof(fsgs_step_plan); }

extern "C" int fsgs_step_forward(fsgs_step_plan *p, fsgs_stream_t stream) {
    if (bad_plan(p)) return FSGS_EINVAL;
    if (p->armed) return FSGS_EPROTOCOL;  // the previous frame's verdict was never looked at
    const int tw = p->tile_width, th = p->tile_height, W = p->width, H = p->height;
    const int64_t P = (int64_t)W * H;
    // the mapped block: total / flag / capacity / bucket overflow (tile_scan.h: bucket_offsets_body256 writes [0], [3], then [1])
    volatile int32_t *m = p->mapped;
    m[1] = 0;
    m[3] = 0;
    m[2] = (int32_t)p->capacity;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    hipStream_t hs = fsgs::as_stream(stream);
    int rc;
    {
    Hook h(p, hs, 0);
    rc = fsgs_project_bin_live_fill_sh_pack(
        p->n, p->means, p->quats, p->log_scales, p->opac_logit, p->binarise, p->binary_threshold, p->viewmat, p->K, W, H, 0.3f,
        0.01f, 1e10f, 0.0f, p->scales_exp, p->opac_sig, p->radii, p->means2d, p->depths, p->conics, tw, th, p->tiles_per_gauss,
        p->isect_offsets, p->bucket_base, p->tile_cursor, p->buckets, p->bucket_words, p->next_bucket_base, p->growth, p->slack,
        p->mapped, p->sh_degree, p->campos, p->features_dc, p->features_rest, p->c2w, p->packed, p->normals_world, p->zero_cells,
        p->n_zero, p->tile_order, stream);
    }
    if (rc) return rc;
    p->armed = 1;
    {
    Hook h(p, hs, 1);
    rc = fsgs_bin_live_sort_buckets(tw, th, p->isect_offsets, p->bucket_base, p->buckets, p->payload, p->long_flag, p->rel_gate,
                                    p->tile_order, stream);
    }
    if (rc) return rc;
    // (n_isects < 0: the lists' true end is isect_offsets[T] on the device, -n_isects the capacity the buffers were sized for)
    {
    Hook h(p, hs, 2);
    rc = fsgs_raster_fwd_quad(1, 4, p->packed, p->payload, p->isect_offsets, -p->capacity, nullptr, W, H, tw, th, 1, p->render,
                              p->alphas, p->last_ids, p->records, p->n_rec, p->seg_state, p->render_extra, p->max_last, nullptr,
                              FSGS_WALK_ONE_WAVE, p->handoff_records > 0 ? p->tail_scratch : nullptr,
                              p->handoff_records > 0 ? p->tail_scratch_bytes : 0, p->handoff_records > 0 ? p->tail_items : 0,
                              p->handoff_records, p->handoff_rel_len, p->long_flag,
                              p->tail_epoch, p->seg_split, p->bwd_queue, p->bwd_queue ? p->bwd_queue_items : 0, p->tile_order,
                              stream);
    }
    if (rc) return rc;
    const bool fusion = p->loss_kind == 1;
    {
    Hook h(p, hs, 3);
    rc = fsgs_epilogue_fwd_order(P, p->render, p->alphas, p->render_extra, p->background, p->max_last, p->n_cells, p->rgb, p->depth,
                                 p->normal, fusion ? nullptr : p->gt_depth, fusion ? nullptr : p->gt_normal,
                                 fusion ? nullptr : p->seed, fusion ? 0.f : p->g_depth, fusion ? 0.f : p->g_normal,
                                 fusion ? nullptr : p->aux_partial, fusion ? nullptr : p->v_depth_img,
                                 (fusion || !p->gt_normal) ? nullptr : p->v_normal_img, p->bwd_order ? p->n_rec : nullptr,
                                 p->bwd_order ? p->seg_split : nullptr, p->bwd_order ? p->order_counters : nullptr, p->bwd_order,
                                 p->bwd_order ? p->order_shift : 0, p->bwd_order ? tw : 0, p->bwd_order ? th : 0, stream);
    }
    if (rc) return rc;
    float *maps = p->ssim_maps;
    const int64_t plane = P * 3;
    if (!fusion) {
        Hook h(p, hs, 4);
        return fsgs_ssim_l1_fwd(H, W, p->rgb, p->gt_rgb, maps, maps + plane, maps + 2 * plane, p->ssim_sums, stream);
    }
    {
    Hook h(p, hs, 4);
    rc = fsgs_ssim_l1_fwd_masked(H, W, p->rgb, p->gt_rgb, p->mask, maps, maps + plane, maps + 2 * plane, p->ssim_sums, stream);
    }
    if (rc) return rc;
    // the depth / normal terms with their gradient images (the seed is known) + the min-scale and touch-normal partial sums
    Hook h(p, hs, 8);
    return fsgs_fusion_aux_loss_riders(H, W, p->depth, p->normal, p->gt_rgb, p->sensor_depth, p->gt_normal, p->mask, p->depth_tol,
                                       p->w_aux, p->seed, p->fa_partial, p->v_depth_img, p->v_normal_img, p->fa_flags,
                                       p->ms_partial ? p->n : 0, p->log_scales, p->ms_partial, p->n_touch, p->touch_idx,
                                       p->normals_world, p->touch_normals, p->touch_partial, stream);
}

extern "C" int fsgs_step_backward(fsgs_step_plan *p, int64_t spin_limit, int64_t *n_live_out, fsgs_stream_t stream) {
    if (bad_plan(p)) return FSGS_EINVAL;
    if (!p->armed) return FSGS_EPROTOCOL;
    const int tw = p->tile_width, th = p->tile_height, W = p->width, H = p->height;
    // the step's one host wait: the flag is raised by launch 1's bookkeeping workgroup, i.e. long before the forward ends
    volatile int32_t *m = p->mapped;
    bool landed = false;
    const auto w0 = std::chrono::steady_clock::now();
    for (int64_t i = 0; i < spin_limit; ++i) {
        if (m[1] != 0) { landed = true; break; }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!landed) {
        if (hipStreamSynchronize(fsgs::as_stream(stream)) != hipSuccess) return fsgs::check_launch();
        if (m[1] == 0) { p->armed = 0; return FSGS_EPROTOCOL; }  // the stream ran dry and the flag never came
    }
    p->wait_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const int64_t total = m[0];
    if (n_live_out) *n_live_out = total;
    p->armed = 0;
    if (total > p->capacity || m[3] != 0) return FSGS_STEP_OVERFLOW;
    const int64_t P = (int64_t)W * H;
    float *maps = p->ssim_maps;
    const int64_t plane = P * 3;
    hipStream_t hs = fsgs::as_stream(stream);
    int rc;
    const bool fusion = p->loss_kind == 1;
    if (!fusion) {
        const float *partials[2] = {p->ssim_sums, p->aux_partial};
        const int64_t rows[2] = {p->ssim_rows, p->aux_rows};
        const float weights[4] = {p->g_l1, p->g_ssim, p->g_depth, p->g_normal};
        Hook h(p, hs, 5);
        rc = fsgs_ssim_l1_bwd_combine(H, W, p->rgb, p->gt_rgb, maps, maps + plane, maps + 2 * plane, p->seed, p->g_l1, p->g_ssim,
                                      p->v_rgb, 2, partials, rows, weights, p->ssim_lambda, p->loss_out, stream);
    } else {
        // every partial-sum column's weight in get_loss_dict's total (ops.fusion_loss_weights), combined by the SSIM backward
        const float *partials[4];
        int64_t rows[4];
        int cols[4];
        float weights[2 + 8 + 2 + 2];
        int n = 0, nw = 0;
        partials[n] = p->ssim_sums; rows[n] = p->ssim_rows; cols[n] = 2; weights[nw++] = p->g_l1; weights[nw++] = p->g_ssim; ++n;
        partials[n] = p->fa_partial; rows[n] = p->fa_rows; cols[n] = 8;
        for (int k = 0; k < 7; ++k) weights[nw++] = p->w_aux[k];
        weights[nw++] = 0.f; ++n;
        if (p->ms_partial) { partials[n] = p->ms_partial; rows[n] = p->ms_rows; cols[n] = 2; weights[nw++] = p->g_min; weights[nw++] = 0.f; ++n; }
        if (p->n_touch > 0) { partials[n] = p->touch_partial; rows[n] = p->touch_rows; cols[n] = 2; weights[nw++] = p->g_touch; weights[nw++] = 0.f; ++n; }
        Hook h(p, hs, 5);
        rc = fsgs_ssim_l1_bwd_masked(H, W, p->rgb, p->gt_rgb, p->mask, maps, maps + plane, maps + 2 * plane, p->seed, p->g_l1,
                                     p->g_ssim, p->v_rgb, n, partials, rows, cols, weights, p->ssim_lambda, p->loss_out, stream);
    }
    if (rc) return rc;
    {
    Hook h(p, hs, 6);
    rc = fsgs_raster_bwd_quad_images(p->records, p->n_rec, p->isect_offsets, p->capacity, W, H, tw, th, p->render, p->alphas,
                                     p->last_ids, p->render_extra, p->background, p->v_rgb, p->v_depth_img,
                                     (fusion || p->gt_normal) ? p->v_normal_img : nullptr, nullptr, p->seg_state, 1, p->v_packed,
                                     p->replica_rows, p->dispatch_stride, p->seg_split, p->bwd_queue,
                                     p->bwd_queue ? p->bwd_queue_items : 0, p->bwd_order, stream);
    }
    if (rc) return rc;
    if (p->adam.n_groups == 0) {
        {
        Hook h(p, hs, 7);
        rc = fsgs_gauss_sh_bwd(p->n, p->sh_degree, p->means, p->campos, p->features_dc, p->features_rest, p->quats, p->log_scales,
                               p->scales_exp, p->opac_sig, p->viewmat, p->K, p->c2w, W, H, 0.3f, p->radii, p->conics, p->v_packed,
                               p->g_features_dc, p->g_features_rest, nullptr, p->g_means, p->g_quats, p->g_log_scales,
                               p->g_opac_logit, p->absgrad, p->xys_grad_norm, p->vis_counts, p->max_2Dsize, p->inv_max_hw,
                               p->frozen, p->replica_rows, p->gsb_flags, stream);
        }
        if (rc || !fusion || p->g_min == 0.f) return rc;
        if (!p->ms_partial) return FSGS_EINVAL;
        // (dn_model.py:817-819: the min-scale term reaches the log-scales directly, touch anchors included)
        return fsgs_min_scale_loss(p->n, p->log_scales, p->g_min, p->seed, p->ms_partial, p->g_log_scales, stream);
    }
    Hook h(p, hs, 7);
    return fsgs_gauss_sh_bwd_adam(p->n, p->sh_degree, p->means, p->campos, p->features_dc, p->features_rest, p->quats,
                                  p->log_scales, p->scales_exp, p->opac_sig, p->viewmat, p->K, p->c2w, W, H, 0.3f, p->radii,
                                  p->conics, p->v_packed, p->absgrad, p->xys_grad_norm, p->vis_counts, p->max_2Dsize,
                                  p->inv_max_hw, p->frozen, p->replica_rows, &p->adam, fusion ? p->g_min : p->min_scale_g, p->gsb_flags, stream);
}

extern "C" int fsgs_step_run(fsgs_step_plan *p, int64_t spin_limit, int64_t *n_live_out, fsgs_stream_t stream) {
    const int rc = fsgs_step_forward(p, stream);
    if (rc) {
        if (p && p->armed) {  // a launch behind the binning failed: the frame is void, leave no armed block behind
            (void)hipStreamSynchronize(fsgs::as_stream(stream));
            p->armed = 0;
        }
        return rc;
    }
    return fsgs_step_backward(p, spin_limit, n_live_out, stream);
}
