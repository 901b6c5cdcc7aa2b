/*
 * meshraster_hip.h -- C-ABI of libmeshraster_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the render + warp hot path of hassony2/handobjectconsist
 * (`meshreg/neurender` + `meshreg/warping`).  The reference reaches its native code
 * through the five pybind11 entry points of `neural_renderer.cuda.rasterize`
 * (imported at /root/reference/meshreg/neurender/rasterize.py:6).  Section 1 below
 * exports exactly those five, with the argument order and the caller-allocates /
 * callee-mutates-in-place contract of the reference call sites.  Sections 2-3 are the
 * fused MI355X-native entry points our own `rasterize.py` / `imgflowarp.py` mirrors
 * use (same results, fewer HBM round trips).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 / int32 memory unless noted;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); nothing in
 *     here synchronises the host, allocates from the host heap or keeps global state;
 *   - return value: 0 = ok, MR_ERR_BADARG (-1) = bad argument, MR_ERR_NOTIMPL (-2),
 *     > 0 = the hipError_t raised by a launch / async allocation;
 *   - raster maps are in RASTER orientation (row 0 = image bottom, the un-flipped
 *     orientation of rasterize.py:443-445) unless the name says `_img`, which means
 *     IMAGE orientation: vertically flipped, NCHW for rgb (rasterize.py:416-428).
 */
#ifndef MESHRASTER_HIP_H
#define MESHRASTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MR_OK 0
#define MR_ERR_BADARG (-1)
#define MR_ERR_NOTIMPL (-2)

/* `flags` bit of the fused entry points: run the straightforward upstream-structured
 * algorithm (every pixel loops over every face / per-pixel global atomics) instead of the
 * tiled one.  Validation and A/B profiling only. */
#define MR_FLAG_REFERENCE_ALGO 1
/* mr_render_flow_forward: tiles of the screen without any candidate face write their coverage bytes (0) and
 * NOTHING else -- the image planes and face_index_map keep whatever they held there.  For callers that consult
 * tile_hit before every read of a rendered plane (mr_occlusion_flow and mr_render_flow_backward do). */
#define MR_FLAG_SPARSE_TILES 2
/* mr_render_flow_backward: grad_vcolors holds zeros on entry (mr_render_flow_forward's zero_fill wrote them, or the
 * caller did): the call does not clear it again -- one launch and 12 B per vertex less on the backward's critical path. */
#define MR_FLAG_OUTPUT_ZEROED 4
/* mr_render_forward / mr_render_vc_forward: one workgroup per screen tile in screen order (rounds 1-3) instead of the
 * tile list + background stream these entry points use since round 4 where the raster allows it.  Same outputs, bit for
 * bit; A/B profiling and tests only. */
#define MR_FLAG_TILE_PER_WORKGROUP 8
/* mr_render_flow_forward: the caller has cleared the first mr_render_clear_bytes(...) bytes at the tile list's header
 * (mr_render_tile_list's list_header, pure address arithmetic on the workspace pointer: the list counters and the arrival
 * counters of the binning pass) on the same stream before the call -- mr_flow_pair_prologue_parts does it on request.  The per-face pass, whose first thread clears it otherwise, then runs inside the binning pass (one launch less). */
#define MR_FLAG_TILE_LIST_CLEARED 16

/* texel_layout argument of the vertex-colour entry points (mr_render_vc_*, mr_render_flow_*): which vertex's colour the
 * three non-zero texels of the 2x2x2 texture of libyana's batch_vertex_textures hold -- two bits per texel axis,
 * sigma(0) | sigma(1) << 2 | sigma(2) << 4, texel (1,0,0) = vertex sigma(0), (0,1,0) = sigma(1), (0,0,1) = sigma(2) of
 * the face.  libyana's source is absent from the reference tree; the default is the identity (ASSUMED).
 * 0 selects the default; anything else must be a permutation of {0,1,2}. */
#define MR_TEXEL_LAYOUT_DEFAULT (0 | 1 << 2 | 2 << 4)

#if defined(__GNUC__)
#define MR_API __attribute__((visibility("default")))
#else
#define MR_API
#endif

typedef void* mr_stream_t;

/* ABI version of this header (bumped on any signature change): what mr_abi_version() of a matching library returns.
 * 2: mr_pair_consist_* coverage arguments, mr_occlusion_flow, mr_render_flow_*, workspace queries (round 2);
 * 3: mr_render_flow_forward tile_bound / tile_count_out / zero_fill, MR_FLAG_OUTPUT_ZEROED, texel_layout of the four
 *    vertex-colour entry points;
 * 4: mr_render_tile_list, mr_occlusion_flow_tiles, mr_pair_consist_{forward,backward}_tiles, mr_pair_consist_tiles_workspace_bytes
 *    (the warp half of the training path over the render's tile list: the sparse contract, round 4),
 *    mr_flow_pair_{forward,backward}_tiles (the same fused into one forward and one backward launch);
 * 5: mr_flow_pair_forward_grad_tiles / mr_flow_pair_backward_unit_tiles (the pair loss's gradient formed by the forward
 *    launch); mr_pair_consist_tiles_workspace_bytes grows to three words per tile;
 * 6: mr_render_clear_bytes; MR_FLAG_TILE_LIST_CLEARED covers that many bytes (list header + the arrival counters of the
 *    binning pass's workgroups, several per image since round 5); mr_flow_pair_prologue_parts takes clear_bytes;
 * 7: scatter_work of mr_flow_pair_forward_grad_tiles / mr_flow_pair_backward_unit_tiles (the covered-tile lists the
 *    backward's workgroups are handed out over), mr_flow_pair_scatter_work_bytes;
 * 8: mr_pair_step_* (the frame-pair step behind one argument struct), mr_pixel_map_terms. */
#define MR_ABI_VERSION 8
MR_API int mr_abi_version(void);
/* 1 if the calling thread's CURRENT HIP device is a gfx950, else 0.
 * Device contract of every entry point below: kernels are launched on the calling thread's current HIP
 * device, on `stream`, which must belong to that device, as must every pointer.  A host with several GPUs
 * selects the device (hipSetDevice) before calling; the Python binding does so from the device of the
 * tensors it passes (handobjectconsist_amd/_lib.py: call). */
MR_API int mr_device_ok(void);
/* Validation only: refined[i] = a[i] / b[i] through the shared-reciprocal sequence the forward tile kernel uses for
 * division-safe faces (csrc/mr_common.hpp: rcp_refined + div_refined), plain[i] = a[i] / b[i] as the compiler divides.
 * Within the operand ranges stated there the two are the same bits (and the CPU's). */
MR_API int mr_selftest_division(const float* a, const float* b, float* refined, float* plain, int64_t n,
                                mr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 1. Upstream-compatible entry points (neural_renderer.cuda.rasterize)
 * ---------------------------------------------------------------------------------- */

/* Replaces rasterize_cuda.forward_face_index_map (rasterize.py:202-215).
 * faces[B,F,3,3] (x,y in NDC, z metric) -> face_index_map[B,is,is] (caller pre-fills -1),
 * weight_map[B,is,is,3] (pre-filled 0), depth_map[B,is,is] (pre-filled `far`),
 * face_inv_map[B,is,is,3,3] (written iff return_depth; else may be a 1-element dummy),
 * faces_inv[B,F,3,3] caller scratch (receives the per-face pixel-space inverse; entries
 * of back-facing faces are left untouched).  Hard z-buffer, lowest face index wins ties. */
MR_API int mr_forward_face_index_map(const float* faces, int32_t* face_index_map, float* weight_map,
                              float* depth_map, float* face_inv_map, float* faces_inv,
                              int batch_size, int num_faces, int image_size, float near_,
                              float far_, int return_rgb, int return_alpha, int return_depth,
                              mr_stream_t stream);

/* Replaces rasterize_cuda.forward_texture_sampling (rasterize.py:232-243).
 * textures[B,F,ts,ts,ts,3]; writes rgb_map[B,is,is,3], sampling_index_map[B,is,is,8],
 * sampling_weight_map[B,is,is,8] at hit pixels only (caller pre-fills 0). */
MR_API int mr_forward_texture_sampling(const float* faces, const float* textures,
                                const int32_t* face_index_map, const float* weight_map,
                                const float* depth_map, float* rgb_map,
                                int32_t* sampling_index_map, float* sampling_weight_map,
                                int batch_size, int num_faces, int image_size,
                                int texture_size, float eps, mr_stream_t stream);

/* Replaces rasterize_cuda.backward_pixel_map (rasterize.py:269-281): the NMR
 * edge-crossing pseudo-gradient.  Writes (does not accumulate) the x,y slots of
 * grad_faces[B,F,3,3] for front-facing faces; z slots and back-facing rows untouched. */
MR_API int mr_backward_pixel_map(const float* faces, const int32_t* face_index_map,
                          const float* rgb_map, const float* alpha_map,
                          const float* grad_rgb_map, const float* grad_alpha_map,
                          float* grad_faces, int batch_size, int num_faces, int image_size,
                          float eps, int return_rgb, int return_alpha, mr_stream_t stream);

/* Replaces rasterize_cuda.backward_textures (rasterize.py:290-297): exact adjoint of
 * the texture sampling; accumulates into grad_textures[B,F,ts,ts,ts,3] (pre-zeroed). */
MR_API int mr_backward_textures(const int32_t* face_index_map, const float* sampling_weight_map,
                         const int32_t* sampling_index_map, const float* grad_rgb_map,
                         float* grad_textures, int batch_size, int num_faces, int image_size,
                         int texture_size, mr_stream_t stream);

/* Replaces rasterize_cuda.backward_depth_map (rasterize.py:306-315): accumulates the
 * analytic d(depth)/d(vertex) into grad_faces[B,F,3,3]. */
MR_API int mr_backward_depth_map(const float* faces, const float* depth_map,
                          const int32_t* face_index_map, const float* face_inv_map,
                          const float* weight_map, const float* grad_depth_map,
                          float* grad_faces, int batch_size, int num_faces, int image_size,
                          mr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 2. Fused MI355X-native render entry points
 * ---------------------------------------------------------------------------------- */

/* Bytes of device workspace mr_render_forward / mr_render_backward need. */
MR_API int64_t mr_render_workspace_bytes(int batch_size, int num_faces, int image_size);

/* Kernels A+B+C + background + alpha + vertical flip + NHWC->NCHW in one pass
 * (= RasterizeFunction.forward rasterize.py:23-125 followed by rasterize_rgbad's
 * permute/flip rasterize.py:413-428).  Every output is fully written (no pre-fill
 * needed).  Any of rgb_img / alpha_img / depth_img / face_inv_map / textures may be
 * NULL when the corresponding return_* flag is 0.
 *   rgb_img[B,3,is,is]  alpha_img[B,is,is]  depth_img[B,is,is]      (IMAGE orientation)
 *   face_index_map[B,is,is] i32  weight_map[B,is,is,3]  face_inv_map[B,is,is,3,3]
 *                                                                  (RASTER orientation)
 * background[3] (bg_stride 0) or [B,3] (bg_stride 3) is a device pointer. */
MR_API int mr_render_forward(const float* faces, const float* textures, const float* background,
                      int bg_stride, float* rgb_img, float* alpha_img, float* depth_img,
                      int32_t* face_index_map, float* weight_map, float* face_inv_map,
                      void* workspace, int64_t workspace_bytes, int batch_size,
                      int num_faces, int image_size, int texture_size, float near_,
                      float far_, float eps, int return_rgb, int return_alpha,
                      int return_depth, int flags, mr_stream_t stream);

/* face_inv_map[B,is,is,3,3] (RASTER orientation) from (faces, face_index_map): the map
 * upstream stores per pixel during the forward pass (rasterize.py:79-83), materialised on
 * demand instead (zeros where no face was hit). */
MR_API int mr_face_inv_map(const float* faces, const int32_t* face_index_map, float* face_inv_map,
                           int batch_size, int num_faces, int image_size, mr_stream_t stream);

/* Kernels E+F (and D when want_grad_faces and (return_rgb or return_alpha)) against the
 * IMAGE-orientation gradients produced by autograd for mr_render_forward's outputs.
 * Sampling indices/weights and the per-pixel inverse are recomputed from
 * (faces, face_index_map) instead of being stored.  grad_faces[B,F,3,3] /
 * grad_textures[B,F,ts,ts,ts,3] are fully written (no pre-zeroing needed); either may
 * be NULL to skip it.  rgb_img / alpha_img are only read by the pixel-map term.
 * workspace: optional scratch of mr_render_backward_workspace_bytes() bytes (37 bytes per face + 20 per image): it
 * holds the flags / the list of the faces that own a pixel -- kernel D and the E / F gather then walk only those (a
 * fifth of a hand + object mesh) -- and, image by image, the owners' pixel-space vertices: with it kernel D runs by
 * strips of image lines staged in LDS (rasters up to 1421 pixels wide; its sums meet in float atomics, so the order
 * of the fp32 additions is not fixed).  NULL / too small / MR_FLAG_REFERENCE_ALGO: the plane-reading per-face walk in
 * upstream's order (same result up to fp32 summation order).  mr_render_backward_list_workspace_bytes() returns the
 * same size (kept for callers of ABI 2, when a call without the pixel-map term needed less). */
MR_API int64_t mr_render_backward_workspace_bytes(int batch_size, int num_faces, int image_size);
MR_API int64_t mr_render_backward_list_workspace_bytes(int batch_size, int num_faces);
MR_API int mr_render_backward(const float* faces, const float* textures,
                       const int32_t* face_index_map, const float* rgb_img,
                       const float* alpha_img, const float* grad_rgb_img,
                       const float* grad_alpha_img, const float* grad_depth_img,
                       float* grad_faces, float* grad_textures, void* workspace,
                       int64_t workspace_bytes, int batch_size, int num_faces,
                       int image_size, int texture_size, float near_, float far_, float eps,
                       int return_rgb, int return_alpha, int return_depth, int flags,
                       mr_stream_t stream);

/* Profiling aid of the pixel-map term (kernel D by strips): mr_render_backward launched with `flags | (1024 << 8)` adds the
 * number of TERMS its walk evaluates -- one term = one evaluation of the sweep body of upstream's backward_pixel_map
 * (rasterize.py:269-281): every position of an "out" sweep, every position of an "in" sweep whose pixel belongs to the face
 * -- to a device-side counter.  This call synchronises the device, copies the counter to *terms_host (host memory) and, with
 * `reset`, clears it.  bench.py prices the launch against terms x 10 lane-instructions (`d_e_f.frac_of_algorithmic_issue`). */
MR_API int mr_pixel_map_terms(uint64_t* terms_host, int reset);

/* Vertex-colour mode (SURVEY 8f "f1"): the render of opticalflow.get_opticalflow, where the
 * face textures are the 2x2x2 vertex-colour textures of batch_vertex_textures
 * (opticalflow.py:101-105) and fill-back doubles the faces (renderer.py:250-252).  Takes the
 * projected vertices verts[B,V,3] (x,y NDC, z metric), the vertex indices faces_idx[B,F0,3]
 * (int32) and the per-vertex colours vcolors[B,V,3] directly: neither the [B,F,3,3] face
 * coordinates, nor the [B,F,2,2,2,3] textures, nor their fill-back copies ever exist in HBM.
 * With fill_back, face index fn >= F0 denotes face fn - F0 with reversed vertex order.
 * Outputs as mr_render_forward (face_index_map values in [0, 2 F0)); bit-identical to rendering
 * the materialised textures.  workspace: mr_render_workspace_bytes(B, fill_back ? 2 F0 : F0, is).
 * texel_layout: see MR_TEXEL_LAYOUT_DEFAULT (0 = default). */
MR_API int mr_render_vc_forward(const float* verts, const int32_t* faces_idx, const float* vcolors,
                                const float* background, int bg_stride, float* rgb_img,
                                float* alpha_img, float* depth_img, int32_t* face_index_map,
                                float* weight_map, void* workspace, int64_t workspace_bytes,
                                int batch_size, int num_verts, int num_faces, int fill_back,
                                int image_size, float near_, float far_, float eps, int return_rgb,
                                int return_alpha, int return_depth, int flags, int texel_layout,
                                mr_stream_t stream);

/* Adjoint of the above w.r.t. vcolors (kernel E composed with the adjoints of
 * batch_vertex_textures and of the fill-back concatenation): grad_vcolors[B,V,3] is zeroed
 * here and accumulated with fp32 atomics.  weight_map (raster orientation) and depth_img
 * (image orientation) are the maps mr_render_vc_forward wrote for the same inputs; both may
 * be NULL, in which case the barycentrics and depths are recomputed from the vertices (same
 * values, more arithmetic). */
MR_API int mr_render_vc_backward(const float* verts, const int32_t* faces_idx,
                                 const int32_t* face_index_map, const float* weight_map,
                                 const float* depth_img, const float* grad_rgb_img,
                                 float* grad_vcolors, int batch_size, int num_verts, int num_faces,
                                 int fill_back, int image_size, float eps, int flags, int texel_layout,
                                 mr_stream_t stream);

/* mr_render_vc_forward restricted to what get_opticalflow consumes from its two renders (opticalflow.py:108-118,
 * 126-135, 146-154) -- the training path's output set: the first two colour planes (the rendered displacement;
 * rgb_img[B,3,is,is], its third plane is left untouched), alpha, and the flow mask of opticalflow.py:109-117
 *   mask = (alpha > alpha_thresh) * keep_lut[face_index + 1]   (keep_lut nullable; 1 beyond n_lut entries)
 * in image orientation, plus face_index_map (raster orientation).  depth_img / weight_map (both nullable) are
 * written at COVERED pixels only -- the loss never reads them and their one reader, mr_render_vc_backward, looks
 * at covered pixels only; everywhere else the buffers keep whatever they held.  tile_hit (nullable): 4 bytes per
 * 32x8 screen tile, [B, ceil(is / 8), ceil(is / 32), 4]; byte w is 1 when rows 2w, 2w + 1 of the tile hold a covered
 * pixel -- what mr_render_flow_backward skips empty tiles on.  flags: MR_FLAG_SPARSE_TILES (requires tile_hit).
 * vertex_id_map (nullable; needs weight_map): [B,is,is,3] int32, raster orientation.  When given, every covered pixel
 * gets the vertex ids of its winning face there and weight_map receives the three SAMPLING weights of the colour
 * taps (the factors the colours of those vertices enter the pixel with) instead of the barycentrics; depth_img may
 * then be NULL.  These per-pixel records are all mr_render_flow_backward needs: one load round trip per pixel
 * instead of face index -> vertex ids -> vertex depths.
 * tile_bound (with MR_FLAG_SPARSE_TILES, rasters of at most 8192 tiles): != 0 makes the binning pass compact the tiles
 * that hold candidate faces into a list -- it writes the zero coverage bytes of all other tiles itself -- and the tile
 * kernel is launched over about tile_bound workgroups that walk this list with a grid stride, instead of one workgroup
 * per screen tile of which four in five find nothing to do.  tile_bound is the caller's GUESS of the list length (> 0;
 * < 0: a quarter of the tiles); any value gives the same images -- a list longer than the grid is walked in several
 * rounds, surplus workgroups leave at once.  tile_count_out (nullable; any address the device can write, e.g. pinned
 * host memory): receives the list length of this call, the natural guess for the next call on similar scenes.
 * tile_bound == 0: one workgroup per tile.
 * zero_fill (nullable) / zero_fill_count: a float buffer the binning pass clears on its way -- meant for the gradient
 * buffer [B, V, 3] of the matching mr_render_flow_backward call (MR_FLAG_OUTPUT_ZEROED there), whose own clearing
 * would sit on the backward pass's critical path. */
MR_API int mr_render_flow_forward(const float* verts, const int32_t* faces_idx, const float* vcolors,
                                  const float* background, int bg_stride, const float* keep_lut, int n_lut,
                                  float alpha_thresh, float* rgb_img, float* alpha_img, float* mask_img,
                                  float* depth_img, float* weight_map, int32_t* face_index_map, uint8_t* tile_hit,
                                  void* workspace, int64_t workspace_bytes,
                                  int batch_size, int num_verts, int num_faces, int fill_back, int image_size,
                                  float near_, float far_, float eps, int flags, int32_t* vertex_id_map,
                                  int tile_bound, uint32_t* tile_count_out, float* zero_fill,
                                  int64_t zero_fill_count, int texel_layout, mr_stream_t stream);

/* Adjoint of mr_render_flow_forward w.r.t. vcolors, with the adjoint of the flow epilogue of get_opticalflow
 * (opticalflow.py:146-154: mask products, permute, [:2], crop) folded in.  The incoming gradient is either
 *   grad_rgb_img[B,3,is,is] (image orientation; then the flow-space arguments are ignored), or, grad_rgb_img NULL,
 *   grad_flow[B,height,width,2] with mask_pre / mask_x / occl [B,is,is] (image orientation):
 *     grad_rgb[b, c, y, x] = (grad_flow[b, y, x, c] * (mask_x * occl)) * mask_pre  inside the crop, 0 outside and
 *     for c = 2 -- never materialised.  mask_x of image b is mask_x_lo + b * is * is for b < split, else
 *     mask_x_hi + (b - split) * is * is (the two directions of a stacked frame pair use different masks).
 * weight_map / depth_img / tile_hit as written by mr_render_flow_forward (tile_hit nullable: every tile is read).
 * vertex_id_map: as written by mr_render_flow_forward together with the sampling weights in weight_map, or NULL
 * (weight_map then holds barycentrics and verts / faces_idx / depth_img are read instead; with the records given those
 * three may be NULL, and texel_layout is not consulted: the records already name the vertices behind the colour taps).
 * grad_bound (nullable; flow-space gradient only): [B] floats, per image an upper bound of |grad_flow| -- e.g. the
 * grad_max mr_pair_consist_backward wrote.  The kernel scales the products into 64-bit fixed point by its workgroup's
 * largest |gradient|; given a bound it does not have to find that maximum in a first pass over its inputs (the masks only
 * shrink the gradient, so any bound of |grad_flow| holds; a bound 2^k too large costs k of the 50 fraction bits).
 * Needs image_size to be a multiple of 4 with at most 4096 tiles, and the [V,3] table to fit LDS (V <= 2560);
 * MR_ERR_NOTIMPL otherwise
 * (callers then use mr_flow_finalize_backward + mr_render_vc_backward). */
MR_API int mr_render_flow_backward(const float* verts, const int32_t* faces_idx, const int32_t* face_index_map,
                                   const uint32_t* tile_hit, const float* weight_map, const float* depth_img,
                                   const float* grad_rgb_img, const float* grad_flow, const float* mask_pre,
                                   const float* mask_x_lo, const float* mask_x_hi, int split, const float* occl,
                                   int height, int width, float* grad_vcolors, int batch_size, int num_verts,
                                   int num_faces, int fill_back, int image_size, float eps, int flags,
                                   const int32_t* vertex_id_map, int texel_layout, const float* grad_bound,
                                   mr_stream_t stream);

/* Per-vertex front end of get_opticalflow in its training setting (SURVEY 8f "f1"): for the two
 * frames of a pair, in one launch
 *   p_k    = batch_proj2d(verts_k, K_k)                      (opticalflow.py:98-99)
 *   cols12 = (p_2 - p_1, 1),  cols21 = (p_1 - p_2, 1)         (opticalflow.py:101-102, 121-122)
 *   ndc_k  = nr.projection(verts_k, K_k, R, t, dist_coeffs, orig_size)   (renderer.py:164-188)
 * verts_k[B,V,3], K_k[B,3,3]; R[Bc,3,3], t[Bc,3], dist_coeffs[Bc,5] with Bc = B if cam_batched
 * else 1.  The outputs feed mr_render_vc_forward (vertices = ndc_k, vcolors = cols). */
MR_API int mr_flow_vertices_forward(const float* verts1, const float* verts2, const float* K1,
                                    const float* K2, const float* R, const float* t,
                                    const float* dist_coeffs, int cam_batched, float orig_size,
                                    float* ndc1, float* ndc2, float* cols12, float* cols21,
                                    int batch_size, int num_verts, mr_stream_t stream);
/* Adjoint of (verts1, verts2) -> (cols12, cols21) (the projected vertices are constants for
 * autograd: detach_renders=True).  Any of grad_cols12 / grad_cols21 / grad_verts1 / grad_verts2
 * may be NULL. */
MR_API int mr_flow_vertices_backward(const float* verts1, const float* verts2, const float* K1,
                                     const float* K2, const float* grad_cols12,
                                     const float* grad_cols21, float* grad_verts1,
                                     float* grad_verts2, int batch_size, int num_verts,
                                     mr_stream_t stream);

/* The same for meshes handed over in TWO parts (hand | object): vertices [0, num_verts_a) of every mesh come from
 * verts*a [B,num_verts_a,3], the rest from verts*b [B,num_verts_b,3] -- the concatenation of warpbranch.py:49-55 done by
 * index instead of by a copy (and a split in the backward).  Outputs as above over V = num_verts_a + num_verts_b; any
 * gradient pointer may be NULL (not wanted). */
MR_API int mr_flow_vertices_parts_forward(const float* verts1a, const float* verts1b, const float* verts2a,
                                          const float* verts2b, int num_verts_a, int num_verts_b, const float* K1,
                                          const float* K2, const float* R, const float* t, const float* dist_coeffs,
                                          int cam_batched, float orig_size, float* ndc1, float* ndc2, float* cols12,
                                          float* cols21, int batch_size, mr_stream_t stream);
MR_API int mr_flow_vertices_parts_backward(const float* verts1a, const float* verts1b, const float* verts2a,
                                           const float* verts2b, int num_verts_a, int num_verts_b, const float* K1,
                                           const float* K2, const float* grad_cols12, const float* grad_cols21,
                                           float* grad_verts1a, float* grad_verts1b, float* grad_verts2a,
                                           float* grad_verts2b, int batch_size, mr_stream_t stream);
/* Faces of the concatenated hand + object mesh of a frame pair as the stacked render takes them: faces_out int32
 * [2B, Fh + Fo, 3], rows b and B + b = hand_faces (shared [Fh,3], or per sample [B,Fh,3] with hand_batched) followed by
 * obj_faces[b] + vertex_offset (warpbranch.py:36, 49-55: repeat, offset, cat) -- one pass instead of five launches. */
MR_API int mr_stack_pair_faces(const int64_t* hand_faces, int hand_batched, const int64_t* obj_faces, int vertex_offset,
                               int32_t* faces_out, int batch_size, int num_hand_faces, int num_obj_faces,
                               mr_stream_t stream);
/* mr_flow_vertices_parts_forward + mr_stack_pair_faces (vertex_offset = num_verts_a) in ONE launch: the two set-up steps of
 * a frame pair do not depend on each other (ABI 5).  clear16 (nullable, 16-byte aligned) / clear_bytes (a
 * multiple of 16): a region the launch clears -- mr_render_clear_bytes at the header of the tile list of the render that
 * follows (MR_FLAG_TILE_LIST_CLEARED). */
MR_API int mr_flow_pair_prologue_parts(const float* verts1a, const float* verts1b, const float* verts2a, const float* verts2b,
                                       int num_verts_a, int num_verts_b, const float* K1, const float* K2, const float* R,
                                       const float* t, const float* dist_coeffs, int cam_batched, float orig_size,
                                       float* ndc1, float* ndc2, float* cols12, float* cols21, const int64_t* hand_faces,
                                       int hand_batched, const int64_t* obj_faces, int32_t* faces_out, int num_hand_faces,
                                       int num_obj_faces, int batch_size, void* clear16, int64_t clear_bytes,
                                       mr_stream_t stream);

/* MANO linear-blend skinning (manopth ManoLayer.forward as called at
 * manobranch.py:130-136, PCA pose space, arithmetic of SURVEY appendix B.10) and its adjoint.
 *   pose_coeffs[B, 3 + ncomps] (global axis-angle + PCA coefficients), betas[B,10]
 *   -> verts_out[B,778,3], jtr_out[B,21,3]: millimetres, centred on joint `center` (or not: -1),
 *      joints = cat(16 chain joints, 5 finger-tip vertices)[reorder].
 * Model constants (device pointers, prepared once by the caller): comps[ncomps,45], hands_mean[45],
 * js[48,10] = J_regressor x shapedirs, jt[48] = J_regressor x template, blend[146,2334] = the shape
 * (rows 0-9) and pose (rows 10-144) blend shapes, coefficient-major, row 145 zero; v_template[2334];
 * weights[778,16]; parents[16] (-1 for the root), tips[5], reorder[21] (int32).
 * The blend-shape product -- the only GEMM-shaped step of the whole path -- runs on the matrix cores
 * (v_mfma_f32_32x32x2_f32).  workspace: mr_mano_workspace_floats(B) floats; mr_mano_backward must get
 * the workspace its forward call filled. */
MR_API int64_t mr_mano_workspace_floats(int batch_size);
MR_API int mr_mano_forward(const float* pose_coeffs, const float* betas, const float* comps,
                           const float* hands_mean, const float* js, const float* jt,
                           const float* blend, const float* v_template, const float* weights,
                           const int32_t* parents, const int32_t* tips, const int32_t* reorder,
                           int ncomps, int center, float* workspace, float* verts_out,
                           float* jtr_out, int batch_size, mr_stream_t stream);
MR_API int mr_mano_backward(const float* comps, const float* hands_mean, const float* js,
                            const float* jt, const float* blend, const float* v_template,
                            const float* weights, const int32_t* parents, const int32_t* tips,
                            const int32_t* reorder, int ncomps, int center, float* workspace,
                            const float* grad_verts, const float* grad_jtr,
                            float* grad_pose_coeffs, float* grad_betas, int batch_size,
                            mr_stream_t stream);

/* The parameter-free geometry between MeshRegNet's regression heads and the render path
 * (recover_3d_proj, project.py:5-24; meshregnet.py:206-245, 274-323; objbranch.py:28-84; libyana
 * batch_proj2d), one launch:
 *   c_h = centre(K, scaletrans)          handverts3d = verts_mm / 1000 + c_h
 *                                        joints3d    = c_h + joints_mm / 1000,  joints2d = proj(K, joints3d)
 *   c_o = centre(K, st_obj[:3]), R = rodrigues(st_obj[3:6])
 *                                        objverts3d  = c_o + R canverts,        objverts2d = proj(K, objverts3d)
 * with centre(K, (s, tx, ty)): Z0 = K00 * s * scale_factor + off_z,
 *   XY0 = ((tx, ty) * trans_factor + (res_w, res_h) / 2 - (K02, K12)) * Z0 / K00.
 * verts_mm[B,Vh,3], joints_mm[B,J,3], scaletrans[B,3], st_obj[B,6], K[B,3,3], canverts[B,Vo,3]. */
MR_API int mr_meshreg_post_forward(const float* verts_mm, const float* joints_mm,
                                   const float* scaletrans, const float* st_obj, const float* K,
                                   const float* canverts, float trans_factor, float scale_factor,
                                   float off_z, float res_w, float res_h, float* handverts3d,
                                   float* joints3d, float* joints2d, float* objverts3d,
                                   float* objverts2d, int batch_size, int num_hand_verts,
                                   int num_joints, int num_obj_verts, mr_stream_t stream);
/* Adjoint w.r.t. verts_mm, joints_mm, scaletrans, st_obj (any output gradient may be NULL);
 * workspace: 15 * batch_size floats. */
MR_API int mr_meshreg_post_backward(const float* verts_mm, const float* joints_mm,
                                    const float* scaletrans, const float* st_obj, const float* K,
                                    const float* canverts, float trans_factor, float scale_factor,
                                    float off_z, float res_w, float res_h,
                                    const float* grad_handverts3d, const float* grad_joints3d,
                                    const float* grad_joints2d, const float* grad_objverts3d,
                                    const float* grad_objverts2d, float* workspace,
                                    float* grad_verts_mm, float* grad_joints_mm,
                                    float* grad_scaletrans, float* grad_st_obj, int batch_size,
                                    int num_hand_verts, int num_joints, int num_obj_verts,
                                    mr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 3. Warping (meshreg/warping/imgflowarp.py)
 * ---------------------------------------------------------------------------------- */

/* imgflowarp.warp (imgflowarp.py:31-55): x[B,C,H,W], flow[B,2,H,W] (pixel units) ->
 * out[B,C,H,W] = sample(x) * mask, mask[B,C,H,W] in {0,1}.  mode 0 = bilinear,
 * 1 = nearest; zeros padding; normalisation by (W-1),(H-1) sampled with
 * align_corners=False. */
MR_API int mr_warp_forward(const float* x, const float* flow, float* out, float* mask,
                    int batch_size, int channels, int height, int width, float thresh,
                    int mode, mr_stream_t stream);

/* Adjoint of mr_warp_forward w.r.t. x (grad_x, pre-zeroed, accumulated with atomics; may
 * be NULL) and w.r.t. flow (grad_flow[B,2,H,W], fully written; may be NULL; zero for
 * mode 1).  The mask carries no gradient (Q7). */
MR_API int mr_warp_backward(const float* x, const float* flow, const float* grad_out,
                     float* grad_x, float* grad_flow, int batch_size, int channels,
                     int height, int width, float thresh, int mode, mr_stream_t stream);

/* imgflowarp.get_occlusion_mask (imgflowarp.py:118-172), the four chained nearest
 * warps fused: mask_flow{1,2}[B,H,W], flow{12,21}[B,>=2,H,W] with channel stride
 * flow_cstride = H*W and batch stride flow_bstride (elements) -> occl{1,2}[B,H,W].
 * flow12_scale / flow21_scale [B,H,W] (nullable): per-pixel factors applied to the flows on
 * load, so that the masked flows rgb * mask of opticalflow.py:118,135 need not be
 * materialised. */
MR_API int mr_occlusion_mask(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                      const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                      const float* flow21_scale, float* occl1, float* occl2, int batch_size,
                      int height, int width, float distance_thresh, float warp_thresh,
                      mr_stream_t stream);

/* mr_occlusion_mask followed by the flow epilogue of opticalflow.py:146-154 for both directions, in one pass:
 *   flow_out12[b, y, x, c] = (flow12[b, c, y, x] * flow12_scale) * (mask_flow1 * occl1),  c = 0, 1, y < crop_height,
 *   x < crop_width  ([B, crop_height, crop_width, 2]); likewise flow_out21 with mask_flow2 / occl2.
 * (= mr_flow_finalize_forward with mask_pre = the scale and mask_x = the mask the occlusion check used.)
 * tile_hit1 / tile_hit2 (both or neither; square rasters): the coverage bytes mr_render_flow_forward wrote for the
 * images behind (mask_flow1, flow12) and (mask_flow2, flow21); where a byte is 0 the planes are not read (they
 * count as background), which is what allows the render to run with MR_FLAG_SPARSE_TILES. */
MR_API int mr_occlusion_flow(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                             const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                             const float* flow21_scale, float* occl1, float* occl2, float* flow_out12,
                             float* flow_out21, const uint8_t* tile_hit1, const uint8_t* tile_hit2, int batch_size,
                             int height, int width, int crop_height, int crop_width, float distance_thresh,
                             float warp_thresh, mr_stream_t stream);

/* Flow epilogue of opticalflow.get_opticalflow (opticalflow.py:109-154), fused.
 * mr_flow_mask: mask[B,is,is] (IMAGE orientation) = (alpha_img > thresh) * keep, where keep
 *   looks the un-flipped face_index_map up in keep_lut[n_lut] (entry f+1 for face f, entry 0 =
 *   background; faces beyond the table are kept) -- the ignore-face mask of :110-116 incl. its
 *   manual vertical flip.  keep_lut may be NULL (no ignore list).
 * mr_flow_finalize_forward: flow[B,H,W,2] = (rgb_img[:, c] * mask_pre) * (mask_x * occl), c = 0,1,
 *   cropped to the top-left H x W (:146-154); all maps [B,is,is] in IMAGE orientation.
 * mr_flow_finalize_backward: its adjoint w.r.t. rgb_img; grad_rgb_img[B,3,is,is] fully written. */
MR_API int mr_flow_mask(const float* alpha_img, const int32_t* face_index_map, const float* keep_lut,
                        int n_lut, float thresh, float* mask, int batch_size, int image_size,
                        mr_stream_t stream);
MR_API int mr_flow_finalize_forward(const float* rgb_img, const float* mask_pre, const float* mask_x,
                                    const float* occl, float* flow, int batch_size, int image_size,
                                    int height, int width, mr_stream_t stream);
MR_API int mr_flow_finalize_backward(const float* grad_flow, const float* mask_pre,
                                     const float* mask_x, const float* occl, float* grad_rgb_img,
                                     int batch_size, int image_size, int height, int width,
                                     mr_stream_t stream);

/* Bytes of device workspace mr_pair_consist_forward needs (per-block partial sums). */
MR_API int64_t mr_pair_consist_workspace_bytes(int batch_size, int height, int width);

/* imgflowarp.pair_consist (imgflowarp.py:58-115) with criterion l1 / level_nb 1
 * (pyramidloss.py:56-62, lossutils.py:1-8), both directions fused into one pass.
 *   flow12, flow21 [B,H,W,2]   image_ref, image [B,3,H,W]   jitter_ref, jitter [B,Cj,H,W],
 *   Cj = jitter_channels in {1, 3}
 * Outputs:
 *   sums[B,4] = {sum1, cnt1, sum2, cnt2}  (masked L1 sums / 3*valid counts; kept for backward)
 *   loss_fwd[B], loss_bwd[B]              sum / (cnt == 0 ? 1 : cnt); warp_loss = fwd (+ bwd)
 * Optional debug outputs of the reference's `masks, warps, diffs` (any may be NULL):
 *   full_mask1/2[B,H,W] u8    warp_mask1/2[B,3,H,W]    warp1/2[B,3,H,W]    diff1/2[B,3,H,W]
 * The reduction is two-stage and deterministic (no float atomics).  Requires width >= 2 and
 * height * width <= 2^29 (the two taps of a row are fetched with one 8-byte load at a 32-bit
 * byte offset); smaller images go through mr_warp_forward.
 * tile_hit12 / tile_hit21 (optional, both or neither): the coverage bytes [B, tiles_y, tiles_x, 4] that
 * mr_render_flow_forward wrote for the renders behind flow12 / flow21 (raster side hit_image_size >= height, width;
 * the flows are the top-left height x width crop of the image-oriented raster).  A rendered flow is exactly zero
 * where its render covered nothing -- such pixels add nothing to the sums -- and with the bytes given it is not even
 * read there (the flows stay fully defined tensors; NULL, NULL, 0 = read everything). */
MR_API int mr_pair_consist_forward(const float* flow12, const float* flow21, const float* image_ref,
                            const float* image, const float* jitter_ref, const float* jitter,
                            int jitter_channels, void* workspace, int64_t workspace_bytes,
                            float* sums, float* loss_fwd, float* loss_bwd, uint8_t* full_mask1,
                            uint8_t* full_mask2, float* warp_mask1, float* warp_mask2,
                            float* warp1, float* warp2, float* diff1, float* diff2,
                            int batch_size, int height, int width, float thresh,
                            const uint8_t* tile_hit12, const uint8_t* tile_hit21, int hit_image_size,
                            mr_stream_t stream);

/* Adjoint of the pair loss w.r.t. the two flows (the only differentiable inputs on the
 * training path): grad_flow12/21[B,H,W,2] fully written (zeros where the coverage bytes say so).  grad_loss_fwd/bwd[B] are the
 * incoming gradients of loss_fwd / loss_bwd (grad_loss_bwd may be NULL).
 * grad_max (nullable): [2 B] floats, ZERO on entry; on return [b] holds max |grad_flow12[b]| and [B + b] max
 * |grad_flow21[b]| (NaN / Inf if any) -- the grad_bound mr_render_flow_backward accepts for the stacked flows. */
MR_API int mr_pair_consist_backward(const float* flow12, const float* flow21, const float* image_ref,
                             const float* image, const float* jitter_ref, const float* jitter,
                             int jitter_channels, const float* sums,
                             const float* grad_loss_fwd, const float* grad_loss_bwd,
                             float* grad_flow12, float* grad_flow21, int batch_size,
                             int height, int width, float thresh, const uint8_t* tile_hit12,
                             const uint8_t* tile_hit21, int hit_image_size, float* grad_max, mr_stream_t stream);

/* ---- the warp half of the training path over the render's tile list (the SPARSE contract) -----------------------
 * The stacked render of a frame pair (mr_render_flow_forward over 2B images: frame 1 of every pair, then frame 2; called
 * with MR_FLAG_SPARSE_TILES and tile_bound != 0) leaves in its workspace the list of the 32 x 8 screen tiles that hold
 * candidate faces.  mr_render_tile_list returns where (device pointers into `workspace`, valid until the workspace is
 * reused; MR_ERR_NOTIMPL for raster sizes that build no list):
 *   list_header: two counters {n_heavy, n_light};  list_entries: uint4[2 * capacity], entry.x = image * tiles + tile
 *   (heavy entries at [0, n_heavy), light ones at [capacity, capacity + n_light));  capacity = batch * tiles.
 * The three *_tiles entry points below are mr_occlusion_flow / mr_pair_consist_forward / mr_pair_consist_backward of that
 * stacked pair launched over the list -- one workgroup per (image of the stack, tile), nothing dispatched for the five
 * sixths of the screen no mesh touches -- with SPARSE outputs:
 *   a tile whose 4-byte coverage word (tile_hit) is non-zero gets every pixel of its outputs written (zeros where
 *   nothing is to be computed); under all other tiles the output buffers keep WHATEVER THEY HELD.
 * Legal for callers that consult the coverage bytes before every read of those buffers: the three entry points do so
 * among themselves, and so does mr_render_flow_backward for grad_flow / occl.  Callers that hand the tensors to anyone else
 * use the dense entry points above.  Arguments are those of the dense entry points (batch_size = B pairs; *1 / *12 =
 * frame 1's grid, images [0, B) of the stack; *2 / *21 = frame 2's, images [B, 2B)) plus the list (capacity must equal
 * 2 B tiles) and tile_bound = the caller's guess of the list length, as for mr_render_flow_forward (any value gives
 * the same results; <= 0: a quarter of the tiles).
 * mr_pair_consist_forward_tiles: workspace of mr_pair_consist_tiles_workspace_bytes(B, hit_image_size) bytes; the
 * reduction is two-stage and deterministic (per-tile partials in fixed order), but its order differs from the dense
 * entry point's (32 x 8 instead of 64 x 4 blocks): sums agree to fp32 rounding, not bit for bit.  No debug outputs. */
MR_API int mr_render_tile_list(const void* workspace, int batch_size, int num_faces, int image_size,
                               const void** list_header, const void** list_entries, int64_t* list_capacity);
/* Bytes at list_header a caller of MR_FLAG_TILE_LIST_CLEARED has to clear (a multiple of 16; sizes only, no device). */
MR_API int64_t mr_render_clear_bytes(int batch_size, int num_faces, int image_size);
MR_API int mr_occlusion_flow_tiles(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                                   const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                                   const float* flow21_scale, float* occl1, float* occl2, float* flow_out12,
                                   float* flow_out21, const uint8_t* tile_hit1, const uint8_t* tile_hit2, int batch_size,
                                   int image_size, int crop_height, int crop_width, float distance_thresh,
                                   float warp_thresh, const void* list_header, const void* list_entries,
                                   int64_t list_capacity, int64_t tile_bound, mr_stream_t stream);
MR_API int64_t mr_pair_consist_tiles_workspace_bytes(int batch_size, int hit_image_size);
MR_API int mr_pair_consist_forward_tiles(const float* flow12, const float* flow21, const float* image_ref,
                                         const float* image, const float* jitter_ref, const float* jitter,
                                         int jitter_channels, void* workspace, int64_t workspace_bytes, float* sums,
                                         float* loss_fwd, float* loss_bwd, int batch_size, int height, int width,
                                         float thresh, const uint8_t* tile_hit12, const uint8_t* tile_hit21,
                                         int hit_image_size, const void* list_header, const void* list_entries,
                                         int64_t list_capacity, int64_t tile_bound, mr_stream_t stream);
MR_API int mr_pair_consist_backward_tiles(const float* flow12, const float* flow21, const float* image_ref,
                                          const float* image, const float* jitter_ref, const float* jitter,
                                          int jitter_channels, const float* sums, const float* grad_loss_fwd,
                                          const float* grad_loss_bwd, float* grad_flow12, float* grad_flow21,
                                          int batch_size, int height, int width, float thresh,
                                          const uint8_t* tile_hit12, const uint8_t* tile_hit21, int hit_image_size,
                                          float* grad_max, const void* list_header, const void* list_entries,
                                          int64_t list_capacity, int64_t tile_bound, mr_stream_t stream);

/* The consistency term of a frame pair in two launches (+ the render).
 * mr_flow_pair_forward_tiles = mr_occlusion_flow_tiles + mr_pair_consist_forward_tiles in ONE pass over the tile list: the
 * thread that has formed its pixel's final flow warps the image with it.  Arguments: those of mr_occlusion_flow_tiles
 * (crop = height x width = the images' size) followed by those of mr_pair_consist_forward_tiles; outputs occl1 / occl2,
 * flow_out12 / flow_out21 (sparse contract) and sums / loss_fwd / loss_bwd -- bit-identical to the two calls.
 * mr_flow_pair_backward_tiles = mr_pair_consist_backward_tiles + mr_render_flow_backward (flow-space form, per-pixel
 * records) in ONE launch over the stacked pair (batch_size = 2B images, split = B): every workgroup computes the pair
 * loss's flow gradient of its tiles itself, times the epilogue masks, and scatters it to grad_vcolors [2B,V,3]; the flow
 * gradient never exists as a tensor.  flows [2B,height,width,2]: the stacked final flows the forward wrote;
 * grad_flow_scratch [2B,height,width,2]: scratch of the launch (contents unspecified afterwards); mask_pre / mask_x_lo /
 * mask_x_hi / occl as for mr_render_flow_backward; sums / grad_loss_fwd / grad_loss_bwd as for mr_pair_consist_backward
 * (grad_loss_bwd nullable); flags: MR_FLAG_OUTPUT_ZEROED.  Same products as the two calls; the fixed-point scale of the
 * per-workgroup sums comes from the workgroup's own largest gradient. */
MR_API int mr_flow_pair_forward_tiles(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                                      const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                                      const float* flow21_scale, float* occl1, float* occl2, float* flow_out12,
                                      float* flow_out21, const uint8_t* tile_hit1, const uint8_t* tile_hit2,
                                      const float* image_ref, const float* image, const float* jitter_ref,
                                      const float* jitter, int jitter_channels, void* workspace, int64_t workspace_bytes,
                                      float* sums, float* loss_fwd, float* loss_bwd, int batch_size, int image_size,
                                      int height, int width, float distance_thresh, float warp_thresh, float pair_thresh,
                                      const void* list_header, const void* list_entries, int64_t list_capacity,
                                      int64_t tile_bound, mr_stream_t stream);
MR_API int mr_flow_pair_backward_tiles(const int32_t* face_index_map, const uint32_t* tile_hit, const float* weight_map,
                                       const int32_t* vertex_id_map, const float* flows, const float* image_ref,
                                       const float* image, const float* jitter_ref, const float* jitter,
                                       int jitter_channels, const float* sums, const float* grad_loss_fwd,
                                       const float* grad_loss_bwd, const float* mask_pre, const float* mask_x_lo,
                                       const float* mask_x_hi, const float* occl, float* grad_flow_scratch, int height,
                                       int width, float* grad_vcolors, int batch_size, int num_verts, int num_faces,
                                       int fill_back, int image_size, float eps, float pair_thresh, int flags,
                                       int texel_layout, mr_stream_t stream);

/* The same pair with the pair loss's gradient formed where its inputs already sit in registers -- in the FORWARD launch
 * (round 4, ABI 5): the training path of opticalflow.flow_pair_loss when the vertices want a gradient.
 * mr_flow_pair_forward_grad_tiles = mr_flow_pair_forward_tiles (same arguments, same outputs bit for bit) + two outputs:
 *   unit_grad [2B,height,width,2]: per pixel of a covered tile, d(sum of |residuals| of its direction) / d(final flow)
 *     -- mr_pair_consist_backward_tiles' gradient for a coefficient grad_loss / count of 1 -- times the epilogue's factors
 *     ((g * (mask_x * occl)) * mask_pre, as mr_render_flow_backward applies them): the gradient w.r.t. the RENDERED
 *     displacement planes up to one scalar per image (sparse contract: untouched under uncovered tiles);
 *   unit_grad_max [2B]: largest |component| of unit_grad per stack image (float; NaN / Inf propagate as such);
 *   loss_sum [B] (nullable): loss_bwd + loss_fwd, pair_consist's warp_loss with use_backward (one element-wise launch less).
 *   workspace: mr_pair_consist_tiles_workspace_bytes(B, image_size) as for the plain call.
 * mr_flow_pair_backward_unit_tiles = the scatter half of mr_flow_pair_backward_tiles on that gradient: every covered pixel's
 *   unit_grad x (grad_loss_{fwd,bwd}[b] / count of the image's direction, from sums) goes to the vertex colours behind its
 *   three sampling weights (per-pixel records of mr_render_flow_forward); no image, mask or flow is read again.  The
 *   fixed-point scale of the per-workgroup sums is unit_grad_max[b] x |coefficient| (an upper bound, rounding is monotone).
 *   A zero coefficient leaves the image's rows zero whatever unit_grad holds.  flags: MR_FLAG_OUTPUT_ZEROED.
 * Results equal mr_flow_pair_backward_tiles' up to fp32 rounding (the coefficient multiplies last instead of first).
 * scatter_work (ABI 7, nullable in both calls; mr_flow_pair_scatter_work_bytes(B, image_size) bytes, contents need no
 *   initialisation): the forward call's finalize launch leaves, per stack image, the number and the ids of its covered tiles
 *   there -- it reads every coverage word anyway; the backward call, given the SAME buffer, hands out its workgroups over the
 *   images in proportion to their covered tiles instead of a fixed number per image (no listing pass; a third of the images of
 *   a hand + object pair otherwise need a second round of their waves while the others' idle).  Null in the backward call:
 *   the listing form (a fixed number of workgroups per image, each compacting the image's coverage words itself).  Same
 *   sums either way up to the order of the final fp32 atomics. */
MR_API int64_t mr_flow_pair_scatter_work_bytes(int batch_size, int image_size);
MR_API int mr_flow_pair_forward_grad_tiles(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                                           const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                                           const float* flow21_scale, float* occl1, float* occl2, float* flow_out12,
                                           float* flow_out21, const uint8_t* tile_hit1, const uint8_t* tile_hit2,
                                           const float* image_ref, const float* image, const float* jitter_ref,
                                           const float* jitter, int jitter_channels, void* workspace, int64_t workspace_bytes,
                                           float* sums, float* loss_fwd, float* loss_bwd, int batch_size, int image_size,
                                           int height, int width, float distance_thresh, float warp_thresh, float pair_thresh,
                                           const void* list_header, const void* list_entries, int64_t list_capacity,
                                           int64_t tile_bound, float* unit_grad, float* unit_grad_max, float* loss_sum,
                                           void* scatter_work, mr_stream_t stream);
MR_API int mr_flow_pair_backward_unit_tiles(const int32_t* face_index_map, const uint32_t* tile_hit, const float* weight_map,
                                            const int32_t* vertex_id_map, const float* unit_grad, const float* unit_grad_max,
                                            const float* sums, const float* grad_loss_fwd, const float* grad_loss_bwd,
                                            int height, int width, float* grad_vcolors, int batch_size, int num_verts,
                                            int num_faces, int fill_back, int image_size, float eps, int flags,
                                            int texel_layout, const void* scatter_work, mr_stream_t stream);

/* ---- ABI 8: the frame-pair step of the training path behind ONE argument block ---------------------------------------
 * What opticalflow.flow_pair_loss issues per frame pair -- mr_flow_pair_prologue_parts, mr_render_flow_forward
 * (MR_FLAG_SPARSE_TILES | MR_FLAG_TILE_LIST_CLEARED, per-pixel records), mr_flow_pair_forward_grad_tiles and, in the backward
 * pass, mr_flow_pair_backward_unit_tiles + mr_flow_vertices_parts_backward -- as TWO calls that take a pointer to one struct
 * (round 5's five calls marshalled 30 - 60 scalars each through the caller's FFI: 0.47 - 0.55 ms of host time per pass for
 * 0.17 ms of device work).  The struct is plain data: a caller fills the size fields once per shape, the pointers per call.
 * Same kernels, same results as the five calls (opticalflow.py:51-156 + imgflowarp.py:58-115 + pyramidloss.py:56-62 +
 * lossutils.py:1-8 for one pair with detach_renders=True), plus the mean over the batch that warpbranch.py:87-88 takes.
 *   scratch: mr_pair_step_sizes' scratch_bytes; everything the forward's launches hand to one another (projected vertices,
 *     stacked faces, the render's workspace and image planes, the occlusion maps, per-tile partials).  Nothing is read from
 *     it after mr_pair_step_forward's launches: ONE buffer per stream can serve every call.
 *   saved: saved_bytes; what mr_pair_step_backward reads (face index map, coverage words at tile_hit_offset -- [2B,
 *     ceil(is/8), ceil(is/32), 4] bytes --, sampling weights, vertex ids, unit gradient and its bounds, per-sample sums,
 *     covered-tile lists) and the gradient buffer the render's binning pass clears: one per differentiated forward call.
 *   flows [2B,height,width,2]: flow12 then flow21, defined under covered tiles only (sparse contract above).
 *   losses [4B + 1]: loss_fwd[B] | loss_bwd[B] | loss_bwd + loss_fwd [B] | mean over the batch of the third block
 *     (mean_of = 0) or of the first (mean_of = 1: pair_consist without use_backward) | B words of scratch (the samples'
 *     values on their way to the workgroup that forms the mean).
 *   backward: grad_loss_fwd / grad_loss_bwd / grad_loss_sum [B] and grad_mean [1] are the incoming gradients of the four
 *     blocks of `losses`, each nullable; the coefficient of a sample's direction is their sum (grad_mean / B for the mean).
 *     grad_verts* nullable (not wanted); all four NULL: nothing to do.  The vertex-colour gradient buffer inside `saved` is
 *     cleared by the FORWARD call (want_grad != 0) and consumed by the first backward call; a caller that differentiates the
 *     same forward call again sets MR_PAIR_STEP_GRAD_BUFFER_USED in `flags` for the later calls (the buffer is then cleared
 *     first).  flags bits 8 and up: profiling switches of the render (forward call only). */
#define MR_PAIR_STEP_GRAD_BUFFER_USED 1
/* flags bit 1 (forward call): every stage as a launch of its own, as in ABI 8's first form -- the vertex stage + stacked faces
 * (mr_flow_pair_prologue_parts, which also clears the list header) otherwise run inside the render's binning pass.  Same
 * results bit for bit; for tests and profiling. */
#define MR_PAIR_STEP_SEPARATE_LAUNCHES 2
/* flags bit 2 (forward call): the caller vouches that the tile-list header inside `scratch` is clean -- `scratch` was last
 * used by a mr_pair_step_forward call of the same sizes that returned MR_OK (its last launch re-zeroes the header), on this
 * stream or one ordered before it.  Without the bit the call clears the header first (one more launch).  A scratch buffer that
 * was never used, was written by anybody else or saw a failed call is NOT clean. */
#define MR_PAIR_STEP_LIST_CLEAN 4
typedef struct MrPairStep {
    int32_t batch_size, num_verts_a, num_verts_b, num_hand_faces, num_obj_faces, hand_faces_batched;
    int32_t fill_back, image_size, height, width, jitter_channels, cam_batched;
    int32_t n_lut, bg_stride, texel_layout, want_grad, mean_of, flags;
    float orig_size, near_, far_, eps, alpha_thresh, distance_thresh, warp_thresh, pair_thresh;
    const float *verts1a, *verts1b, *verts2a, *verts2b, *K1, *K2, *R, *t, *dist_coeffs;
    const int64_t *hand_faces, *obj_faces;
    const float *keep_lut, *background, *image_ref, *image, *jitter_ref, *jitter;
    void *scratch, *saved;
    int64_t scratch_bytes, saved_bytes;
    float *flows, *losses;
    uint32_t* tile_count_out;
    int64_t tile_bound;
    const float *grad_loss_fwd, *grad_loss_bwd, *grad_loss_sum, *grad_mean;
    float *grad_verts1a, *grad_verts1b, *grad_verts2a, *grad_verts2b;
} MrPairStep;
/* sizeof(MrPairStep) and the byte offset of every field in declaration order (at most `capacity` written; returns the
 * number of fields): what a binding checks its own struct definition against. */
MR_API int64_t mr_pair_step_struct_bytes(void);
MR_API int mr_pair_step_field_offsets(int64_t* offsets, int capacity);
/* Buffer sizes for the size fields of *step (pointers are not looked at); MR_ERR_NOTIMPL where the fused path does not apply
 * (no tile list for this raster, image_size % 4, more than 2560 vertices). */
MR_API int mr_pair_step_sizes(const MrPairStep* step, int64_t* scratch_bytes, int64_t* saved_bytes, int64_t* tile_hit_offset);
MR_API int mr_pair_step_forward(const MrPairStep* step, mr_stream_t stream);
MR_API int mr_pair_step_backward(const MrPairStep* step, mr_stream_t stream);

/* ---- dataset pipeline: decoded frames -> network-input batch ------------------------------
 * One launch for a whole batch of what meshreg/datasets/handobjset.py:361-379 does per sample on the
 * host: transform_img (libyana -> PIL Image.transform(size, AFFINE, coeffs), NEAREST, zero fill), crop to
 * (width, height), torchvision to_tensor (u8 / 255) and normalize ((x - mean) / std; the reference uses
 * mean 0.5, std 1), and the jitter mask = the same transform of an all-white image (:361-362, :376-378),
 * after the optional left-right flip of handobjset.py:124-125.
 *   frames[N, src_height, src_width, 3] u8 (HWC, as decoded); coeffs[N,6] f64 = Pillow's (a,b,c,d,e,f):
 *   output pixel (x, y) <- input pixel (a x + b y + c, d x + e y + f); flip[N] u8 or NULL.
 *   image[N,3,height,width] f32; jittermask[N,mask_channels,height,width] f32 in {0,1} or NULL
 *   (mask_channels 3 = the reference's batch format, 1 = one plane).
 * Source pixels are selected exactly as Pillow's Geometry.c does (scale / 16.16 fixed-point / double
 * regimes): bit-exact with the reference's host path.  Non-finite coefficients: the frame is empty.
 * workspace: mr_frames_to_batch_workspace_bytes(N, height, width) bytes, 16-byte aligned. */
MR_API int64_t mr_frames_to_batch_workspace_bytes(int num_frames, int height, int width);
MR_API int mr_frames_to_batch(const uint8_t* frames, const double* coeffs, const uint8_t* flip,
                              float mean0, float mean1, float mean2, float std0, float std1,
                              float std2, void* workspace, int64_t workspace_bytes, float* image,
                              float* jittermask, int mask_channels, int num_frames, int src_height,
                              int src_width, int height, int width, mr_stream_t stream);

/* ---- trainer side: BatchNorm with frozen statistics + residual add + ReLU -----------------
 * The reference trains with --freeze_batchnorm (trainmeshwarp.py:205-206, 237-240): every BatchNorm2d of the
 * ResNet-18 trunk runs in eval mode with trainable affine parameters, followed by ReLU, by "+ identity, ReLU"
 * or by nothing (resnet.py:46-58, 31-43).  One pass each way instead of 2-3 element-wise kernels forward and
 * batch_norm_backward + threshold_backward:
 *   z = (x - running_mean) * (weight / sqrt(running_var + eps)) + bias [+ residual];  y = relu ? max(z, 0) : z
 * x, residual, y, grad_x, grad_y, grad_residual: [batch_size, channels, plane] contiguous (NCHW, plane = H * W)
 * of the activation type act_dtype: 0 = fp32, 1 = bf16 (the trunk under bf16 autocast: converted on load, fp32
 * arithmetic, rounded to nearest-even on store); the channel arrays and their gradients are fp32 [channels].
 * channels_last = 1: the activations are in NHWC memory order [batch_size, plane, channels] (what MIOpen's
 * convolutions prefer); requires channels to be a power of two in [4, 1024] and 16-byte (bf16: 8-byte) alignment.  Backward: g = relu && !(z > 0) ? 0 : grad_y;  grad_x = g * weight / sqrt(var + eps);
 * grad_residual = g (NULL if not wanted); grad_bias = sum g; grad_weight = sum g * (x - mean) / sqrt(var + eps)
 * (either may be NULL; two-stage deterministic reduction through the workspace).  grad_y2 (nullable): a second
 * gradient of y, added to grad_y on load -- y usually feeds two consumers (the next block's convolution and its
 * identity branch), and summing the two gradients here saves autograd's separate add pass. */
MR_API int mr_bn_act_forward(const void* x, const void* residual, const float* weight, const float* bias,
                             const float* running_mean, const float* running_var, float eps, int relu,
                             int act_dtype, int channels_last, void* y, int batch_size, int channels,
                             int plane, mr_stream_t stream);
MR_API int64_t mr_bn_act_backward_workspace_bytes(int batch_size, int channels);
MR_API int mr_bn_act_backward(const void* grad_y, const void* grad_y2, const void* x, const void* residual,
                              const float* weight, const float* bias, const float* running_mean,
                              const float* running_var, float eps, int relu, int act_dtype,
                              int channels_last, void* grad_x, void* grad_residual, float* grad_weight,
                              float* grad_bias,
                              void* workspace, int64_t workspace_bytes, int batch_size, int channels,
                              int plane, mr_stream_t stream);

/* The ResNet stem after its 7x7 convolution as one kernel each way (resnet.py:140-147 with frozen statistics):
 *   y = MaxPool2d(kernel 3, stride 2, padding 1)(relu(bn(x)))     x[N,C,H,W] -> y[N,C,(H-1)/2+1,(W-1)/2+1]
 * The full-resolution activation is never written; the backward recomputes it per tile, re-derives every
 * window's arg-max with PyTorch's rule (kh, kw ascending, strictly greater wins, padding skipped) and gathers the
 * pooled gradient per input pixel (no atomics).  x, y, grad_x, grad_y are of act_dtype (0 = fp32, 1 = bf16);
 * grad_weight / grad_bias (fp32) may be NULL.  channels_last = 1: NHWC memory order; the forward then also writes
 * argmax[N,OH,OW,C] (u8, position kh * 3 + kw of every pooled value) which the backward reads instead of
 * re-deriving it (argmax is unused and may be NULL for NCHW); channels must be a power of two in [4, 1024]. */
MR_API int mr_stem_pool_forward(const void* x, const float* weight, const float* bias,
                                const float* running_mean, const float* running_var, float eps,
                                int act_dtype, int channels_last, void* y, unsigned char* argmax,
                                int batch_size, int channels, int height, int width, mr_stream_t stream);
MR_API int64_t mr_stem_pool_backward_workspace_bytes(int batch_size, int channels, int height, int width);
MR_API int mr_stem_pool_backward(const void* grad_y, const void* grad_y2, const void* x,
                                 const unsigned char* argmax,
                                 const float* weight, const float* bias, const float* running_mean,
                                 const float* running_var, float eps, int act_dtype, int channels_last,
                                 void* grad_x, float* grad_weight, float* grad_bias, void* workspace,
                                 int64_t workspace_bytes, int batch_size, int channels, int height,
                                 int width, mr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MESHRASTER_HIP_H */
