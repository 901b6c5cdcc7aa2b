// pair_step.hip -- the frame-pair step of the training path behind ONE argument block (ABI 8; include/meshraster_hip.h).
//
// What /root/reference/meshreg/models/warpbranch.py:59-88 runs per frame pair -- get_opticalflow (opticalflow.py:51-156) with
// detach_textures=False, detach_renders=True, pair_consist (imgflowarp.py:58-115) with PyramidCriterion("l1")
// (pyramidloss.py:56-62, lossutils.py:1-8) and the mean over the batch -- is five launches forward and two backward in this
// library.  Round 5's caller issued them through five C calls that marshalled 30 - 60 scalars each, allocated a dozen buffers
// and kept two autograd nodes: 0.47 - 0.55 ms of host time per pass for 0.17 ms of device work.  Here the caller fills one
// plain struct (sizes once per shape, pointers per call) and makes ONE call each way; the buffers the launches hand to one
// another live in a scratch region the caller keeps from call to call.  No new device code except the batch mean, which the
// finalize launch's last workgroup forms: every launch is the existing entry point's, so the results are the five calls' bit
// for bit.
#include <cstddef>
#include <cstdint>

#include "mr_common.hpp"
#include "vertex_stage_device.hpp"

int mr_flow_pair_backward_unit_tiles_ex(const int32_t* face_index_map, const uint32_t* tile_hit, const float* weight_map,
                                        const int32_t* vertex_id_map, const float* unit_grad, const float* unit_grad_max,
                                        const float* sums, const float* grad_loss_fwd, const float* grad_loss_bwd,
                                        const float* grad_loss_sum, const float* grad_mean, int mean_of, int height, int width,
                                        float* grad_vcolors, int batch_size, int num_verts, int num_faces, int fill_back,
                                        int image_size, float eps, int flags, int texel_layout, const void* scatter_work,
                                        mr_stream_t stream);

int mr_flow_pair_forward_grad_tiles_ex(const float* mask_flow1, const float* mask_flow2, const float* flow12, const float* flow21,
                                       int64_t flow_bstride, const float* flow12_scale, const float* flow21_scale, float* occl1,
                                       float* occl2, float* flow_out12, float* flow_out21, const uint8_t* tile_hit1,
                                       const uint8_t* tile_hit2, const float* image_ref, const float* image, const float* jitter_ref,
                                       const float* jitter, int jitter_channels, void* workspace, int64_t workspace_bytes,
                                       float* sums, float* loss_fwd, float* loss_bwd, int batch_size, int image_size, int height,
                                       int width, float distance_thresh, float warp_thresh, float pair_thresh, const void* list_header,
                                       const void* list_entries, int64_t list_capacity, int64_t tile_bound, float* unit_grad,
                                       float* unit_grad_max, float* loss_sum, void* scatter_work, float* mean_out, int mean_of,
                                       int reset_list, const void* records, mr_stream_t stream);

int mr_render_flow_forward_pair(const float* verts, const int32_t* faces_idx, const float* vcolors, const float* background,
                                int bg_stride, const float* keep_lut, int n_lut, float alpha_thresh, float* rgb_img, float* alpha_img,
                                float* mask_img, float* depth_img, float* weight_map, int32_t* face_index_map, uint8_t* tile_hit,
                                void* workspace, int64_t workspace_bytes, int batch_size, int num_verts, int num_faces, int fill_back,
                                int image_size, float near_, float far_, float eps, int flags, int32_t* vertex_id_map, int tile_bound,
                                uint32_t* tile_count_out, float* zero_fill, int64_t zero_fill_count, int texel_layout,
                                mr_stream_t stream, const mr::PairPrologue* pro, void* records);

namespace mr {

static inline int64_t ps_align(int64_t x) { return (x + 255) & ~(int64_t)255; }

// where everything sits inside the two buffers (byte offsets, 256-byte aligned)
struct PairStepLayout {
    // scratch
    int64_t ndc, cols, faces2, rgb, alpha, mask, occl, rec, render_work, pair_work, scratch_total;
    int64_t render_work_bytes, pair_work_bytes;
    // saved
    int64_t fim, tile_hit, wmap, vid, unit_grad, unit_max, sums, scatter_work, grad_buf, saved_total;
    int64_t scatter_work_bytes;
    int F0, F, V, B2;
};

static int pair_step_layout(const MrPairStep& a, PairStepLayout& L) {
    if (a.batch_size < 0 || a.num_verts_a < 0 || a.num_verts_b <= 0 || a.num_hand_faces < 0 || a.num_obj_faces < 0 ||
        a.image_size <= 0 || a.height <= 0 || a.width < 2 || a.height > a.image_size || a.width > a.image_size ||
        (a.jitter_channels != 1 && a.jitter_channels != 3) || a.batch_size > (1 << 20))
        return MR_ERR_BADARG;
    const int64_t B2 = 2LL * a.batch_size, V = (int64_t)a.num_verts_a + a.num_verts_b;
    const int64_t F0 = (int64_t)a.num_hand_faces + a.num_obj_faces, F = a.fill_back ? 2 * F0 : F0;
    const int64_t is = a.image_size, px = is * is;
    if (V > 0x7fffffffLL || F > 0x7fffffffLL || B2 * px > (1LL << 40)) return MR_ERR_BADARG;
    L.F0 = (int)F0; L.F = (int)F; L.V = (int)V; L.B2 = (int)B2;
    // the fused path's conditions (opticalflow._stacked_flow_node_ok + has_tile_list)
    if (is % 4 != 0 || ((is + 31) / 32) * ((is + 7) / 8) > 4096 || V > 2560) return MR_ERR_NOTIMPL;
    const void *hdr = nullptr, *ents = nullptr;
    int64_t cap = 0;
    if (a.batch_size > 0 && mr_render_tile_list((const void*)256, (int)B2, (int)F, (int)is, &hdr, &ents, &cap) != MR_OK) return MR_ERR_NOTIMPL;
    L.render_work_bytes = mr_render_workspace_bytes((int)B2, (int)F, (int)is);
    L.pair_work_bytes = mr_pair_consist_tiles_workspace_bytes(a.batch_size, (int)is);
    L.scatter_work_bytes = mr_flow_pair_scatter_work_bytes(a.batch_size, (int)is);
    if (L.render_work_bytes < 0 || L.pair_work_bytes < 0 || L.scatter_work_bytes < 0) return MR_ERR_BADARG;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o += ps_align(bytes > 16 ? bytes : 16); return at; };
    L.ndc = take(B2 * V * 12); L.cols = take(B2 * V * 12); L.faces2 = take(B2 * F0 * 12);
    L.rgb = take(B2 * 3 * px * 4); L.alpha = take(B2 * px * 4); L.mask = take(B2 * px * 4); L.occl = take(B2 * px * 4);
    L.rec = take(B2 * px * 16);  // (round 6: the render's 16-byte pixel records; the planes above serve MR_PAIR_STEP_SEPARATE_LAUNCHES)
    L.render_work = take(L.render_work_bytes); L.pair_work = take(L.pair_work_bytes);
    L.scratch_total = o;
    o = 0;
    const int64_t tiles = ((is + 7) / 8) * ((is + 31) / 32);
    L.fim = take(B2 * px * 4); L.tile_hit = take(B2 * tiles * 4); L.wmap = take(B2 * px * 12); L.vid = take(B2 * px * 12);
    L.unit_grad = take(B2 * (int64_t)a.height * a.width * 8); L.unit_max = take(B2 * 4); L.sums = take((int64_t)a.batch_size * 16);
    L.scatter_work = take(L.scatter_work_bytes); L.grad_buf = take(B2 * V * 12);
    L.saved_total = o;
    return MR_OK;
}

}  // namespace mr

using namespace mr;

extern "C" int64_t mr_pair_step_struct_bytes(void) { return (int64_t)sizeof(MrPairStep); }

extern "C" int mr_pair_step_field_offsets(int64_t* offsets, int capacity) {
    if (!offsets || capacity < 0) return MR_ERR_BADARG;
#define MR_PS_FIELDS(X)                                                                                                        \
    X(batch_size) X(num_verts_a) X(num_verts_b) X(num_hand_faces) X(num_obj_faces) X(hand_faces_batched) X(fill_back)          \
    X(image_size) X(height) X(width) X(jitter_channels) X(cam_batched) X(n_lut) X(bg_stride) X(texel_layout) X(want_grad)     \
    X(mean_of) X(flags) X(orig_size) X(near_) X(far_) X(eps) X(alpha_thresh) X(distance_thresh) X(warp_thresh) X(pair_thresh) \
    X(verts1a) X(verts1b) X(verts2a) X(verts2b) X(K1) X(K2) X(R) X(t) X(dist_coeffs) X(hand_faces) X(obj_faces) X(keep_lut)    \
    X(background) X(image_ref) X(image) X(jitter_ref) X(jitter) X(scratch) X(saved) X(scratch_bytes) X(saved_bytes) X(flows)   \
    X(losses) X(tile_count_out) X(tile_bound) X(grad_loss_fwd) X(grad_loss_bwd) X(grad_loss_sum) X(grad_mean) X(grad_verts1a) \
    X(grad_verts1b) X(grad_verts2a) X(grad_verts2b)
    int n = 0;
#define MR_PS_ONE(f) if (n < capacity) offsets[n] = (int64_t)offsetof(MrPairStep, f); n++;
    MR_PS_FIELDS(MR_PS_ONE)
#undef MR_PS_ONE
#undef MR_PS_FIELDS
    return n;
}

extern "C" int mr_pair_step_sizes(const MrPairStep* step, int64_t* scratch_bytes, int64_t* saved_bytes, int64_t* tile_hit_offset) {
    if (!step) return MR_ERR_BADARG;
    PairStepLayout L;
    const int rc = pair_step_layout(*step, L);
    if (rc != MR_OK) return rc;
    if (scratch_bytes) *scratch_bytes = L.scratch_total;
    if (saved_bytes) *saved_bytes = L.saved_total;
    if (tile_hit_offset) *tile_hit_offset = L.tile_hit;
    return MR_OK;
}

extern "C" int mr_pair_step_forward(const MrPairStep* step, mr_stream_t stream) {
    if (!step) return MR_ERR_BADARG;
    const MrPairStep& a = *step;
    PairStepLayout L;
    int rc = pair_step_layout(a, L);
    if (rc != MR_OK) return rc;
    if (a.batch_size == 0) return MR_OK;
    if (!a.verts1a || !a.verts1b || !a.verts2a || !a.verts2b || !a.K1 || !a.K2 || !a.R || !a.t || !a.dist_coeffs || !a.hand_faces ||
        !a.obj_faces || !a.background || !a.image_ref || !a.image || !a.jitter_ref || !a.jitter || !a.scratch || !a.saved ||
        !a.flows || !a.losses || a.scratch_bytes < L.scratch_total || a.saved_bytes < L.saved_total ||
        ((uintptr_t)a.scratch & 15) || ((uintptr_t)a.saved & 15) || !(a.eps >= 1e-6f))
        return MR_ERR_BADARG;
    char* sc = (char*)a.scratch;
    char* sv = (char*)a.saved;
    const int B = a.batch_size, B2 = L.B2, V = L.V, is = a.image_size;
    const int64_t px = (int64_t)is * is;
    float* ndc = (float*)(sc + L.ndc);
    float* cols = (float*)(sc + L.cols);
    int32_t* faces2 = (int32_t*)(sc + L.faces2);
    float *rgb = (float*)(sc + L.rgb), *alpha = (float*)(sc + L.alpha), *mask = (float*)(sc + L.mask), *occl = (float*)(sc + L.occl);
    void* rwork = sc + L.render_work;
    void* pwork = sc + L.pair_work;
    int32_t* fim = (int32_t*)(sv + L.fim);
    uint8_t* tile_hit = (uint8_t*)(sv + L.tile_hit);
    float* wmap = (float*)(sv + L.wmap);
    int32_t* vid = (int32_t*)(sv + L.vid);
    float *unit_grad = (float*)(sv + L.unit_grad), *unit_max = (float*)(sv + L.unit_max), *sums = (float*)(sv + L.sums);
    void* swork = sv + L.scatter_work;
    float* grad_buf = a.want_grad ? (float*)(sv + L.grad_buf) : nullptr;
    const void *hdr = nullptr, *ents = nullptr;
    int64_t cap = 0;
    rc = mr_render_tile_list(rwork, B2, L.F, is, &hdr, &ents, &cap);
    if (rc != MR_OK) return rc;
    const int64_t clear_bytes = mr_render_clear_bytes(B2, L.F, is);
    // 1. vertex stage of both frames + the stacked faces + the clear of the render's list header.  Round 6: no launch of its
    // own -- the vertex stage and the faces ride in the render's binning pass (bin_boxes_kernel PROLOGUE, raster_fwd.hip), and
    // the list header is left clean by the finalize launch of the PREVIOUS step on this scratch (MR_PAIR_STEP_LIST_CLEAN; a
    // caller that cannot vouch for that gets a memset in front).  MR_PAIR_STEP_SEPARATE_LAUNCHES: the launch of ABI 8's first form.
    const bool separate = (a.flags & MR_PAIR_STEP_SEPARATE_LAUNCHES) != 0;
    PairPrologue pro{};
    if (separate) {
        rc = mr_flow_pair_prologue_parts(a.verts1a, a.verts1b, a.verts2a, a.verts2b, a.num_verts_a, a.num_verts_b, a.K1, a.K2, a.R, a.t,
                                         a.dist_coeffs, a.cam_batched, a.orig_size, ndc, ndc + (int64_t)B * V * 3, cols,
                                         cols + (int64_t)B * V * 3, a.hand_faces, a.hand_faces_batched, a.obj_faces, faces2,
                                         a.num_hand_faces, a.num_obj_faces, B, const_cast<void*>(hdr), clear_bytes, stream);
        if (rc != MR_OK) return rc;
    } else {
        if (!(a.orig_size > 0.0f) || B > 65535) return MR_ERR_BADARG;
        if (!(a.flags & MR_PAIR_STEP_LIST_CLEAN)) {
            const hipError_t e = hipMemsetAsync(const_cast<void*>(hdr), 0, (size_t)clear_bytes, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
        pro.v = VertexStageParams{a.verts1a, a.verts2a, a.verts1b, a.verts2b, a.num_verts_a, a.K1, a.K2, a.R, a.t, a.dist_coeffs,
                                  a.cam_batched ? 1 : 0, a.orig_size, ndc, ndc + (int64_t)B * V * 3, cols, cols + (int64_t)B * V * 3, B, V};
        pro.f = StackFacesParams{a.hand_faces, a.hand_faces_batched ? (int64_t)a.num_hand_faces * 3 : (int64_t)0, a.obj_faces,
                                 a.num_verts_a, faces2, B, a.num_hand_faces, a.num_obj_faces};
    }
    // 2. the flow-mode render of the 2B stacked meshes (binning pass with the per-face pass inside + tile kernel)
    int64_t bound = a.tile_bound;
    if (bound == 0) bound = -1;
    // (round 6: what the render hands to the fused warp forward is ONE 16-byte record per pixel {displacement x, y, alpha,
    // mask} instead of four planes; the first form keeps the planes)
    void* records = separate ? nullptr : (void*)(sc + L.rec);
    rc = mr_render_flow_forward_pair(ndc, faces2, cols, a.background, a.bg_stride, a.keep_lut, a.n_lut, a.alpha_thresh,
                                     records ? nullptr : rgb, records ? nullptr : alpha, records ? nullptr : mask,
                                     nullptr, wmap, fim, tile_hit, rwork, L.render_work_bytes, B2, V, L.F0, a.fill_back, is, a.near_, a.far_,
                                     a.eps, MR_FLAG_SPARSE_TILES | MR_FLAG_TILE_LIST_CLEARED | (a.flags & ~0xff), vid,  // (flags >> 8: the render's profiling switches)
                                     (int)(bound > 0x7fffffffLL ? 0x7fffffffLL : bound), a.tile_count_out, grad_buf,
                                     grad_buf ? (int64_t)B2 * V * 3 : 0, a.texel_layout, stream, separate ? nullptr : &pro, records);
    if (rc != MR_OK) return rc;
    // 3. occlusion + flow epilogue + pair loss forward (+ its unit gradient) over the render's tile list; finalize
    float *loss_fwd = a.losses, *loss_bwd = a.losses + B, *loss_sum = a.losses + 2 * (int64_t)B;
    const int64_t th_half = (int64_t)B * (((is + 7) / 8) * ((is + 31) / 32)) * 4;  // (bytes of frame 1's coverage words)
    float* flow12 = a.flows;
    float* flow21 = a.flows + (int64_t)B * a.height * a.width * 2;
    rc = mr_flow_pair_forward_grad_tiles_ex(mask, alpha + (int64_t)B * px, rgb, rgb + (int64_t)B * 3 * px, 3 * px, mask,
                                         mask + (int64_t)B * px, records ? nullptr : occl, records ? nullptr : occl + (int64_t)B * px, flow12, flow21, tile_hit,
                                         tile_hit + th_half, a.image_ref, a.image, a.jitter_ref, a.jitter, a.jitter_channels, pwork,
                                         L.pair_work_bytes, sums, loss_fwd, loss_bwd, B, is, a.height, a.width, a.distance_thresh,
                                         a.warp_thresh, a.pair_thresh, hdr, ents, cap, bound, unit_grad, unit_max, loss_sum, swork,
                                         // (the finalize launch's last workgroup leaves the mean in losses[3 B]; the B words
                                         // behind it -- the caller's buffer has 4 B + 1 -- carry the samples' values to it)
                                         // (reset_list: the finalize launch leaves the list header's counters zero for the next step)
                                         a.losses + 3 * (int64_t)B, a.mean_of, 1, records, stream);
    return rc;
}

extern "C" int mr_pair_step_backward(const MrPairStep* step, mr_stream_t stream) {
    if (!step) return MR_ERR_BADARG;
    const MrPairStep& a = *step;
    PairStepLayout L;
    int rc = pair_step_layout(a, L);
    if (rc != MR_OK) return rc;
    if (a.batch_size == 0) return MR_OK;
    if (!a.grad_verts1a && !a.grad_verts1b && !a.grad_verts2a && !a.grad_verts2b) return MR_OK;
    if (!a.want_grad || !a.saved || a.saved_bytes < L.saved_total || !a.verts1a || !a.verts1b || !a.verts2a || !a.verts2b || !a.K1 ||
        !a.K2 || !(a.grad_loss_fwd || a.grad_loss_bwd || a.grad_loss_sum || a.grad_mean))
        return MR_ERR_BADARG;
    char* sv = (char*)a.saved;
    const int B = a.batch_size, B2 = L.B2, V = L.V, is = a.image_size;
    float* grad_cols = (float*)(sv + L.grad_buf);
    // (a direction without its own gradient array still takes the sum's / the mean's share: the scatter adds them up)
    rc = mr_flow_pair_backward_unit_tiles_ex((const int32_t*)(sv + L.fim), (const uint32_t*)(sv + L.tile_hit), (const float*)(sv + L.wmap),
                                             (const int32_t*)(sv + L.vid), (const float*)(sv + L.unit_grad),
                                             (const float*)(sv + L.unit_max), (const float*)(sv + L.sums), a.grad_loss_fwd,
                                             a.grad_loss_bwd, a.grad_loss_sum, a.grad_mean, a.mean_of, a.height, a.width, grad_cols, B2,
                                             V, L.F0, a.fill_back, is, a.eps, (a.flags & MR_PAIR_STEP_GRAD_BUFFER_USED) ? 0 : MR_FLAG_OUTPUT_ZEROED, a.texel_layout,
                                             sv + L.scatter_work, stream);
    if (rc != MR_OK) return rc;
    return mr_flow_vertices_parts_backward(a.verts1a, a.verts1b, a.verts2a, a.verts2b, a.num_verts_a, a.num_verts_b, a.K1, a.K2,
                                           grad_cols, grad_cols + (int64_t)B * V * 3, a.grad_verts1a, a.grad_verts1b, a.grad_verts2a,
                                           a.grad_verts2b, B, stream);
}
