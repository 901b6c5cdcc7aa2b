// warp.hip -- photometric warp kernels for gfx950 (MI355X).
//
// Replaces the PyTorch op chains of /root/reference/meshreg/warping/imgflowarp.py:
//   warp (:31-55)                  -> warp_forward_kernel / warp_backward_kernel
//   get_occlusion_mask (:118-172)  -> occlusion_kernel (four chained nearest warps fused into
//                                     two dependent gathers per pixel, no intermediates)
//   pair_consist (:58-115) + PyramidCriterion('l1').compute (pyramidloss.py:56-62) +
//   batch_masked_mean_loss (lossutils.py:1-8)
//                                  -> pair_consist_forward_kernel (both directions, 8 grid
//                                     samples, mask algebra and the masked L1 sums in ONE pass
//                                     over the pixels) + finalize + backward w.r.t. the flows.
//   opticalflow.py:109-154 mask algebra / crop / permute
//                                  -> flow_mask_kernel, flow_finalize_{forward,backward}_kernel
// All are HBM-streaming kernels: one lane per pixel, channel planes read coalesced, the
// bilinear taps hit L1/L2 (flows are a few pixels); in the pair kernels the two taps of a row
// come from ONE 8-byte load at a 32-bit offset from a wave-uniform plane pointer (load-instruction
// count is what bounds them).  Reductions are two-stage and deterministic (fixed-order block
// partials -> per-sample finalize), no float atomics.
//
// grid_sample semantics restated (torch, zeros padding, align_corners=False):
//   vx = 2 (x + u) / max(W - 1, 1) - 1 ;  ix = ((vx + 1) W - 1) / 2        
#include <algorithm>

#include "mr_common.hpp"
#include "warp_device.hpp"

namespace mr {

// ---------------------------------------------------------------------------------------
// warp
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) warp_forward_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ flow,
                                                           float* __restrict__ out, float* __restrict__ mask,
                                                           int B, int C, int H, int W, float thresh, int mode) {
    const int64_t hw = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * hw) return;
    const int b = (int)(i / hw);
    const int64_t pix = i % hw;
    const int yy = (int)(pix / W), xx = (int)(pix % W);
    const float u = flow[((int64_t)b * 2 + 0) * hw + pix], v = flow[((int64_t)b * 2 + 1) * hw + pix];
    float ix, iy;
    sample_pos((float)xx, (float)yy, u, v, W, H, ix, iy);
    if (mode == 0) {
        const Taps t = make_taps(ix, iy);
        const float m = valid_mask(t, W, H, thresh);
        for (int c = 0; c < C; c++) {
            const int64_t o = ((int64_t)b * C + c) * hw;
            out[o + pix] = bilin(x + o, t, W, H) * m;
            mask[o + pix] = m;
        }
    } else {
        int xn, yn;
        nearest_idx(ix, iy, xn, yn);
        const bool ok = inb(xn, yn, W, H);
        float m = ok ? 1.0f : 0.0f;
        if (m < thresh) m = 0.0f;
        if (m > 0.0f) m = 1.0f;
        for (int c = 0; c < C; c++) {
            const int64_t o = ((int64_t)b * C + c) * hw;
            const float s = ok ? x[o + (int64_t)yn * W + xn] : 0.0f;
            out[o + pix] = s * m;
            mask[o + pix] = m;
        }
    }
}

__global__ void __launch_bounds__(256) warp_backward_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ flow,
                                                            const float* __restrict__ grad_out,
                                                            float* __restrict__ grad_x,
                                                            float* __restrict__ grad_flow, int B, int C, int H,
                                                            int W, float thresh, int mode) {
    const int64_t hw = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * hw) return;
    const int b = (int)(i / hw);
    const int64_t pix = i % hw;
    const int yy = (int)(pix / W), xx = (int)(pix % W);
    const float u = flow[((int64_t)b * 2 + 0) * hw + pix], v = flow[((int64_t)b * 2 + 1) * hw + pix];
    float ix, iy;
    sample_pos((float)xx, (float)yy, u, v, W, H, ix, iy);
    float gu = 0.0f, gv = 0.0f;
    if (mode == 0) {
        const Taps t = make_taps(ix, iy);
        const float m = valid_mask(t, W, H, thresh);
        for (int c = 0; c < C; c++) {
            const int64_t o = ((int64_t)b * C + c) * hw;
            const float g = grad_out[o + pix] * m;
            if (grad_flow) {
                float gix, giy;
                bilin_grad(x + o, t, W, H, gix, giy);
                gu += g * gix;
                gv += g * giy;
            }
            if (grad_x && g != 0.0f) {
                if (inb(t.x0, t.y0, W, H)) atomicAdd(&grad_x[o + (int64_t)t.y0 * W + t.x0], g * t.nw);
                if (inb(t.x0 + 1, t.y0, W, H)) atomicAdd(&grad_x[o + (int64_t)t.y0 * W + t.x0 + 1], g * t.ne);
                if (inb(t.x0, t.y0 + 1, W, H)) atomicAdd(&grad_x[o + (int64_t)(t.y0 + 1) * W + t.x0], g * t.sw);
                if (inb(t.x0 + 1, t.y0 + 1, W, H))
                    atomicAdd(&grad_x[o + (int64_t)(t.y0 + 1) * W + t.x0 + 1], g * t.se);
            }
        }
        // d ix / d u = (W / 2) * (2 / max(W - 1, 1))
        gu = gu * ((float)W / 2.0f) * (2.0f / (float)max(W - 1, 1));
        gv = gv * ((float)H / 2.0f) * (2.0f / (float)max(H - 1, 1));
    } else if (grad_x) {
        int xn, yn;
        nearest_idx(ix, iy, xn, yn);
        if (inb(xn, yn, W, H)) {
            float m = 1.0f;
            if (m < thresh) m = 0.0f;
            for (int c = 0; c < C; c++) {
                const int64_t o = ((int64_t)b * C + c) * hw;
                const float g = grad_out[o + pix] * m;
                if (g != 0.0f) atomicAdd(&grad_x[o + (int64_t)yn * W + xn], g);
            }
        }
    }
    if (grad_flow) {
        grad_flow[((int64_t)b * 2 + 0) * hw + pix] = gu;
        grad_flow[((int64_t)b * 2 + 1) * hw + pix] = gv;
    }
}

// ---------------------------------------------------------------------------------------
// forward-backward occlusion check
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) occlusion_kernel(const float* __restrict__ mask1,
                                                        const float* __restrict__ mask2,
                                                        const float* __restrict__ flow12,
                                                        const float* __restrict__ flow21, int64_t fbstride,
                                                        const float* __restrict__ scale12,
                                                        const float* __restrict__ scale21,
                                                        float* __restrict__ occl1, float* __restrict__ occl2,
                                                        int B, int H, int W, float dist_thresh, float wthresh) {
    const int64_t hw = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * hw) return;
    const int b = (int)(i / hw);
    const int64_t pix = i % hw;
    const int yy = (int)(pix / W), xx = (int)(pix % W);
    const float* m1 = mask1 + (int64_t)b * hw;
    const float* m2 = mask2 + (int64_t)b * hw;
    const float* f12 = flow12 + (int64_t)b * fbstride;
    const float* f21 = flow21 + (int64_t)b * fbstride;
    const float* s12 = scale12 ? scale12 + (int64_t)b * hw : nullptr;
    const float* s21 = scale21 ? scale21 + (int64_t)b * hw : nullptr;
    occl1[i] = occl_one(m1, m2, f12, f21, s12, s21, hw, H, W, xx, yy, dist_thresh, wthresh);
    occl2[i] = occl_one(m2, m1, f21, f12, s21, s12, hw, H, W, xx, yy, dist_thresh, wthresh);
}

// occlusion_kernel + flow_finalize_forward_kernel of both directions in one pass (mr_occlusion_flow): the pixel
// that has just computed its two occlusion bits also holds everything the final flows need --
// out12[b, y, x, c] = (flow12[b, c, y, x] * scale12) * (mask1 * occl1), likewise 21 -- inside the crop.
__global__ void __launch_bounds__(256) occlusion_flow_kernel(const float* __restrict__ mask1,
                                                             const float* __restrict__ mask2,
                                                             const float* __restrict__ flow12,
                                                             const float* __restrict__ flow21, int64_t fbstride,
                                                             const float* __restrict__ scale12,
                                                             const float* __restrict__ scale21,
                                                             float* __restrict__ occl1, float* __restrict__ occl2,
                                                             float* __restrict__ out12, float* __restrict__ out21,
                                                             int B, int H, int W, int crop_h, int crop_w,
                                                             float dist_thresh, float wthresh,
                                                             const uint8_t* __restrict__ hit1,
                                                             const uint8_t* __restrict__ hit2, int tiles_x,
                                                             int tiles_y) {
    const int64_t hw = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * hw) return;
    const int b = (int)(i / hw);
    const int64_t pix = i % hw;
    const int yy = (int)(pix / W), xx = (int)(pix % W);
    const float* m1 = mask1 + (int64_t)b * hw;
    const float* m2 = mask2 + (int64_t)b * hw;
    const float* f12 = flow12 + (int64_t)b * fbstride;
    const float* f21 = flow21 + (int64_t)b * fbstride;
    const float* s12 = scale12 ? scale12 + (int64_t)b * hw : nullptr;
    const float* s21 = scale21 ? scale21 + (int64_t)b * hw : nullptr;
    const uint8_t* h1 = hit1 ? hit1 + (int64_t)b * tiles_x * tiles_y * 4 : nullptr;
    const uint8_t* h2 = hit2 ? hit2 + (int64_t)b * tiles_x * tiles_y * 4 : nullptr;
    const float o1 = occl_one(m1, m2, f12, f21, s12, s21, hw, H, W, xx, yy, dist_thresh, wthresh, h1, h2, tiles_x);
    const float o2 = occl_one(m2, m1, f21, f12, s21, s12, hw, H, W, xx, yy, dist_thresh, wthresh, h2, h1, tiles_x);
    occl1[i] = o1;
    occl2[i] = o2;
    if (yy < crop_h && xx < crop_w) {
        const int64_t o = ((int64_t)b * crop_h + yy) * crop_w + xx;
        // (an occlusion bit can only be non-zero at a pixel inside its own image's coverage: the guarded reads of
        // occl_one came first)
        const float a1 = (o1 != 0.0f && s12) ? s12[pix] : 1.0f, a2 = (o2 != 0.0f && s21) ? s21[pix] : 1.0f;
        const float post1 = o1 != 0.0f ? m1[pix] * o1 : 0.0f, post2 = o2 != 0.0f ? m2[pix] * o2 : 0.0f;
        // (x * a) * 0 is a zero for the finite rendered values: not loaded where the masks are 0
        float2 r12 = make_float2(0.0f, 0.0f), r21 = make_float2(0.0f, 0.0f);
        if (post1 != 0.0f && a1 != 0.0f) r12 = make_float2((f12[pix] * a1) * post1, (f12[hw + pix] * a1) * post1);
        if (post2 != 0.0f && a2 != 0.0f) r21 = make_float2((f21[pix] * a2) * post2, (f21[hw + pix] * a2) * post2);
        *reinterpret_cast<float2*>(out12 + o * 2) = r12;
        *reinterpret_cast<float2*>(out21 + o * 2) = r21;
    }
}

// ---------------------------------------------------------------------------------------
// flow epilogue of opticalflow.get_opticalflow (opticalflow.py:109-154)
// ---------------------------------------------------------------------------------------
// mask[b, yi_img, x] = (alpha > thresh) * keep(face_index_map[b, is - 1 - yi_img, x])
// keep(f) = lut[f + 1] for f + 1 < n_lut, else 1 (lut[0] is the background slot)
__global__ void __launch_bounds__(256) flow_mask_kernel(const float* __restrict__ alpha,
                                                        const int32_t* __restrict__ fim,
                                                        const float* __restrict__ lut, int n_lut, float thresh,
                                                        float* __restrict__ mask, int64_t npx, int is) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    float m = (alpha[i] > thresh) ? 1.0f : 0.0f;
    if (lut) {
        const int64_t plane = (int64_t)is * is;
        const int64_t b = i / plane;
        const int pix = (int)(i % plane);
        const int yi = pix / is, xi = pix % is;
        const int f = fim[b * plane + (int64_t)(is - 1 - yi) * is + xi] + 1;
        m = m * ((f >= 0 && f < n_lut) ? lut[f] : 1.0f);
    }
    mask[i] = m;
}

// flow[b, y, x, c] = (rgb[b, c, y, x] * m_pre) * (m_x * occl), c = 0, 1, cropped to H x W
__global__ void __launch_bounds__(256) flow_finalize_forward_kernel(
    const float* __restrict__ rgb, const float* __restrict__ m_pre, const float* __restrict__ m_x,
    const float* __restrict__ occl, float* __restrict__ flow, int B, int is, int H, int W) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H * W) return;
    const int b = (int)(i / ((int64_t)H * W));
    const int pix = (int)(i % ((int64_t)H * W));
    const int y = pix / W, x = pix % W;
    const int64_t plane = (int64_t)is * is, o = (int64_t)b * plane + (int64_t)y * is + x;
    const float a = m_pre[o], post = m_x[o] * occl[o];
    const float r0 = rgb[(int64_t)b * 3 * plane + (int64_t)y * is + x], r1 = rgb[((int64_t)b * 3 + 1) * plane + (int64_t)y * is + x];
    *reinterpret_cast<float2*>(flow + i * 2) = make_float2((r0 * a) * post, (r1 * a) * post);
}

// grad_rgb[b, c, y, x] = grad_flow[b, y, x, c] * post * m_pre inside the crop, 0 elsewhere / for c = 2
__global__ void __launch_bounds__(256) flow_finalize_backward_kernel(
    const float* __restrict__ grad_flow, const float* __restrict__ m_pre, const float* __restrict__ m_x,
    const float* __restrict__ occl, float* __restrict__ grad_rgb, int B, int is, int H, int W) {
    const int64_t plane = (int64_t)is * is;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * plane) return;
    const int64_t b = i / plane;
    const int pix = (int)(i % plane);
    const int y = pix / is, x = pix % is;
    float g0 = 0.0f, g1 = 0.0f;
    if (y < H && x < W) {
        const float a = m_pre[i], post = m_x[i] * occl[i];
        const float2 g = *reinterpret_cast<const float2*>(grad_flow + ((b * H + y) * (int64_t)W + x) * 2);
        g0 = (g.x * post) * a;
        g1 = (g.y * post) * a;
    }
    grad_rgb[b * 3 * plane + pix] = g0;
    grad_rgb[(b * 3 + 1) * plane + pix] = g1;
    grad_rgb[(b * 3 + 2) * plane + pix] = 0.0f;
}

// ---------------------------------------------------------------------------------------
// pair_consist
// ---------------------------------------------------------------------------------------
struct PairParams {
    const float* flow12;  // [B,H,W,2]
    const float* flow21;
    const float* image_ref;  // [B,3,H,W]
    const float* image;
    const float* jitter_ref;  // [B,Cj,H,W]
    const float* jitter;
    int Cj;
    float* partial;  // [B, nblk, 4]
    uint8_t* full_mask1;
    uint8_t* full_mask2;
    float* warp_mask1;  // [B,3,H,W]
    float* warp_mask2;
    float* warp1;
    float* warp2;
    float* diff1;
    float* diff2;
    int B, H, W, nblk, tiles_x;
    float thresh;
    // optional coverage bytes of the renders the flows came from (mr_render_flow_forward): a flow is exactly 0
    // where its render covered nothing, and is not even read there
    const uint8_t* hit12;  // [B, tiles_y, tiles_x, 4] of the render behind flow12 (frame 1)
    const uint8_t* hit21;
    int hit_is, hit_tiles_x, hit_stride;  // raster size, tiles per raster row, bytes per image
};

// 64 x 4 pixel tiles (one 256-B row per wave); logical tile id -> (sample, tile) with the XCD-aware remap so that
// vertically adjacent tiles (which share bilinear taps) run behind the same L2.
constexpr int PT_W = 64, PT_H = 4;
// tiles per workgroup of the BACKWARD kernel: once most blocks leave early a launch is bound by its workgroup count (43
// instead of 46 us in the training step).  The forward, with its block reduction per tile, is slower that way (56 us).
constexpr int PT_SUB = 4;
__device__ __forceinline__ bool pair_tile_pixel(unsigned lid, int H, int W, int tiles_x, int ntiles, int& b, int& tile,
                                                int& xx, int& yy) {
    b = lid / ntiles;
    tile = lid % ntiles;
    xx = (tile % tiles_x) * PT_W + (threadIdx.x % PT_W);
    yy = (tile / tiles_x) * PT_H + (threadIdx.x / PT_W);
    return xx < W && yy < H;
}

// Does the render behind a flow cover anything in the block's PT_W x PT_H pixels?  Block-uniform (scalar loads of
// the 4-byte coverage words of the <= 3 x 2 raster tiles the block touches); NULL = no information = yes.
__device__ __forceinline__ bool pair_block_covered(const uint8_t* __restrict__ hit, int hit_is, int hit_tiles_x,
                                                   int hit_stride, int b, int tile, int tiles_x, int H, int W) {
    if (!hit) return true;
    const int x0 = (tile % tiles_x) * PT_W, y0 = (tile / tiles_x) * PT_H;
    const int x1 = min(x0 + PT_W, W) - 1, y1 = min(y0 + PT_H, H) - 1;
    const uint32_t* h32 = reinterpret_cast<const uint32_t*>(hit + (int64_t)b * hit_stride);
    uint32_t any = 0u;
    for (int ty = (hit_is - 1 - y1) >> 3; ty <= (hit_is - 1 - y0) >> 3; ty++)
        for (int tx = x0 >> 5; tx <= x1 >> 5; tx++) any |= h32[ty * hit_tiles_x + tx];
    return any != 0u;
}

// ... in either of two renders: the words of both are requested before any is looked at (one round trip, not two --
// five sixths of the blocks of a hand + object frame pair end here, and their number times this latency over the
// resident workgroups is the floor of the launch)
__device__ __forceinline__ bool pair_block_covered2(const uint8_t* __restrict__ hit_a, const uint8_t* __restrict__ hit_b,
                                                    int hit_is, int hit_tiles_x, int hit_stride, int b, int tile, int tiles_x,
                                                    int H, int W) {
    const int x0 = (tile % tiles_x) * PT_W, y0 = (tile / tiles_x) * PT_H;
    const int x1 = min(x0 + PT_W, W) - 1, y1 = min(y0 + PT_H, H) - 1;
    const uint32_t* a32 = reinterpret_cast<const uint32_t*>(hit_a + (int64_t)b * hit_stride);
    const uint32_t* b32 = reinterpret_cast<const uint32_t*>(hit_b + (int64_t)b * hit_stride);
    uint32_t any = 0u;
    for (int ty = (hit_is - 1 - y1) >> 3; ty <= (hit_is - 1 - y0) >> 3; ty++)
        for (int tx = x0 >> 5; tx <= x1 >> 5; tx++) any |= a32[ty * hit_tiles_x + tx] | b32[ty * hit_tiles_x + tx];
    return any != 0u;
}

// block-wide sums of four values with ONE barrier: wave butterflies, 4 x 4 partials in LDS,
// every thread adds the four wave partials in wave order (fixed order: deterministic)
__device__ __forceinline__ void block_sum4(float* v, float (*red)[4]) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] += __shfl_xor(v[k], off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 4; k++) red[wave][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
}

__global__ void __launch_bounds__(256, 6) pair_consist_forward_kernel(PairParams p) {
    __shared__ float red[4][4];
    const int64_t hw = (int64_t)p.H * p.W;
    int b, tile, xx, yy;
    const bool in_img = pair_tile_pixel(xcd_remap(blockIdx.x, gridDim.x), p.H, p.W, p.tiles_x, p.nblk, b, tile, xx, yy);
    // nothing rendered under this block in either frame: both flows are zero here, no pixel is valid -- the block's
    // partial sums are zero (no loads, no barrier), unless the per-pixel debug outputs want every pixel
    if (p.hit12 && p.hit21 && !(p.warp1 || p.warp2 || p.diff1 || p.diff2 || p.warp_mask1 || p.warp_mask2 ||
                                p.full_mask1 || p.full_mask2) &&
        !pair_block_covered2(p.hit12, p.hit21, p.hit_is, p.hit_tiles_x, p.hit_stride, b, tile, p.tiles_x, p.H, p.W)) {
        if (threadIdx.x < 4) p.partial[((int64_t)b * p.nblk + tile) * 4 + threadIdx.x] = 0.0f;
        return;
    }
    float sum1 = 0.0f, cnt1 = 0.0f, sum2 = 0.0f, cnt2 = 0.0f;
    if (in_img) {
        const int64_t pix = (int64_t)yy * p.W + xx;
        const bool allj = p.warp_mask1 != nullptr || p.warp_mask2 != nullptr;
        // forward term: image_ref warped by flow21 vs image (imgflowarp.py:80,85-87,93-102)
        // backward term: image warped by flow12 vs image_ref (:84,82,88,99-107)
        const float2 uv1 = pair_flow(p.flow21, p.hit21, p.hit_is, p.hit_tiles_x, p.hit_stride, b, xx, yy, p.H, p.W);
        const float2 uv2 = pair_flow(p.flow12, p.hit12, p.hit_is, p.hit_tiles_x, p.hit_stride, b, xx, yy, p.H, p.W);
        // A pixel whose flow has a zero x component is invalid whatever the images hold (imgflowarp.py:93-100,
        // SURVEY Q5) and adds nothing to the masked sums: unless the per-pixel outputs (warps, differences,
        // warp masks) are requested, neither its taps are computed nor its 26 tap loads issued -- a rendered flow is
        // exactly 0 outside the meshes, 90 % of a hand + object frame.
        const bool per_pixel_out = p.warp1 || p.warp2 || p.diff1 || p.diff2 || p.warp_mask1 || p.warp_mask2;
        const bool need1 = per_pixel_out || uv1.x != 0.0f, need2 = per_pixel_out || uv2.x != 0.0f;
        DirTaps t1{}, t2{};
        if (need1) t1 = pair_taps(uv1, xx, yy, p.H, p.W);
        if (need2) t2 = pair_taps(uv2, xx, yy, p.H, p.W);
        DirRaw2 q1{}, q2{};
        if (need1) pair_load(t1, p.image_ref, p.image, p.jitter, p.jitter, p.Cj, allj, b, pix, hw, q1);
        if (need2) pair_load(t2, p.image, p.image_ref, p.jitter_ref, p.jitter_ref, p.Cj, allj, b, pix, hw, q2);
        pin(q1); pin(q2);
        const DirRaw r1 = unpack(q1, t1.a), r2 = unpack(q2, t2.a);
        DirOut d1{}, d2{};
        if (need1) d1 = pair_eval(t1, r1, p.H, p.W, p.thresh, allj && p.Cj == 3);
        if (need2) d2 = pair_eval(t2, r2, p.H, p.W, p.thresh, allj && p.Cj == 3);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int64_t o = ((int64_t)b * 3 + c) * hw + pix;
            const float df = fabsf(d1.s[c] - r1.tgt[c]);
            const float db = fabsf(d2.s[c] - r2.tgt[c]);
            if (d1.valid) sum1 += df;
            if (d2.valid) sum2 += db;
            if (p.warp1) p.warp1[o] = d1.s[c];
            if (p.warp2) p.warp2[o] = d2.s[c];
            if (p.diff1) p.diff1[o] = df;
            if (p.diff2) p.diff2[o] = db;
            if (p.warp_mask1) p.warp_mask1[o] = d1.wm[c];
            if (p.warp_mask2) p.warp_mask2[o] = d2.wm[c];
        }
        if (d1.valid) cnt1 = 3.0f;
        if (d2.valid) cnt2 = 3.0f;
        if (p.full_mask1) p.full_mask1[(int64_t)b * hw + pix] = d1.valid ? 1 : 0;
        if (p.full_mask2) p.full_mask2[(int64_t)b * hw + pix] = d2.valid ? 1 : 0;
    }
    float v[4] = {sum1, cnt1, sum2, cnt2};
    block_sum4(v, red);
    if (threadIdx.x == 0) {
        float* o = p.partial + ((int64_t)b * p.nblk + tile) * 4;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
}

// per-sample fixed-order reduction of the block partials -> sums[B,4], loss_fwd/bwd[B]
__global__ void __launch_bounds__(64) pair_consist_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                                   float* __restrict__ sums,
                                                                   float* __restrict__ loss_fwd,
                                                                   float* __restrict__ loss_bwd) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int j = lane; j < nblk; j += 64) {
        const float4 v = *reinterpret_cast<const float4*>(partial + ((int64_t)b * nblk + j) * 4);
        a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] += __shfl_xor(a[k], off);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) sums[b * 4 + k] = a[k];
        const float n1 = (a[1] == 0.0f) ? 1.0f : a[1], n2 = (a[3] == 0.0f) ? 1.0f : a[3];
        if (loss_fwd) loss_fwd[b] = a[0] / n1;
        if (loss_bwd) loss_bwd[b] = a[2] / n2;
    }
}

struct PairBwdParams {
    const float* flow12;
    const float* flow21;
    const float* image_ref;
    const float* image;
    const float* jitter_ref;
    const float* jitter;
    int Cj;
    const float* sums;  // [B,4]
    const float* grad_loss_fwd;
    const float* grad_loss_bwd;  // nullable
    float* grad_flow12;
    float* grad_flow21;
    int B, H, W, ntiles, tiles_x;
    float thresh;
    const uint8_t* hit12;  // see PairParams
    const uint8_t* hit21;
    int hit_is, hit_tiles_x, hit_stride;
    unsigned* grad_max;    // nullable: [2B] float bits, zero on entry: max |grad_flow12[b]| at [b], |grad_flow21[b]| at [B + b]
};

__global__ void __launch_bounds__(256) pair_consist_backward_kernel(PairBwdParams p) {
    const int64_t hw = (int64_t)p.H * p.W;
    const unsigned total = (unsigned)p.ntiles * (unsigned)p.B;
    const unsigned lid0 = xcd_remap(blockIdx.x, gridDim.x) * PT_SUB;
#pragma unroll 1
    for (unsigned lid = lid0; lid < min(lid0 + PT_SUB, total); lid++) {
    unsigned u12 = 0u, u21 = 0u;  // |gradient| of this thread's pixel as float bits (NaN > Inf > finite: a plain unsigned max)
    [&] {
    int b, tile, xx, yy;
    if (!pair_tile_pixel(lid, p.H, p.W, p.tiles_x, p.ntiles, b, tile, xx, yy)) return;
    const int64_t pix = (int64_t)yy * p.W + xx;
    if (p.hit12 && p.hit21 &&
        !pair_block_covered2(p.hit12, p.hit21, p.hit_is, p.hit_tiles_x, p.hit_stride, b, tile, p.tiles_x, p.H, p.W)) {
        // nothing rendered under this block in either frame: zero gradient, nothing read
        *reinterpret_cast<float2*>(p.grad_flow21 + ((int64_t)b * hw + pix) * 2) = make_float2(0.0f, 0.0f);
        *reinterpret_cast<float2*>(p.grad_flow12 + ((int64_t)b * hw + pix) * 2) = make_float2(0.0f, 0.0f);
        return;
    }
    const float c1 = p.sums[b * 4 + 1], c2 = p.sums[b * 4 + 3];
    const float coef1 = p.grad_loss_fwd[b] / ((c1 == 0.0f) ? 1.0f : c1);
    const float coef2 = p.grad_loss_bwd ? p.grad_loss_bwd[b] / ((c2 == 0.0f) ? 1.0f : c2) : 0.0f;
    const bool both = p.grad_loss_bwd != nullptr;
    const float2 uv1 = pair_flow(p.flow21, p.hit21, p.hit_is, p.hit_tiles_x, p.hit_stride, b, xx, yy, p.H, p.W);
    const float2 uv2 = both ? pair_flow(p.flow12, p.hit12, p.hit_is, p.hit_tiles_x, p.hit_stride, b, xx, yy, p.H, p.W) : uv1;
    // the gradient of an invalid pixel is 0: no taps, no tap loads where the flow's x component is zero (see the
    // forward kernel)
    const bool need1 = uv1.x != 0.0f && coef1 != 0.0f, need2 = both && uv2.x != 0.0f && coef2 != 0.0f;
    DirTaps t1{}, t2{};
    if (need1) t1 = pair_taps(uv1, xx, yy, p.H, p.W);
    if (need2) t2 = pair_taps(uv2, xx, yy, p.H, p.W);
    DirRaw2 q1{}, q2{};
    if (need1) pair_load(t1, p.image_ref, p.image, p.jitter, p.jitter, p.Cj, false, b, pix, hw, q1);
    if (need2) pair_load(t2, p.image, p.image_ref, p.jitter_ref, p.jitter_ref, p.Cj, false, b, pix, hw, q2);
    pin(q1); pin(q2);
    float2 g21 = make_float2(0.0f, 0.0f), g12 = make_float2(0.0f, 0.0f);
    if (need1) {
        const DirRaw r1 = unpack(q1, t1.a);
        const DirOut d1 = pair_eval(t1, r1, p.H, p.W, p.thresh, false);
        g21 = pair_grad(t1, r1, d1, p.H, p.W, coef1);
    }
    if (need2) {
        const DirRaw r2 = unpack(q2, t2.a);
        const DirOut d2 = pair_eval(t2, r2, p.H, p.W, p.thresh, false);
        g12 = pair_grad(t2, r2, d2, p.H, p.W, coef2);
    }
    *reinterpret_cast<float2*>(p.grad_flow21 + ((int64_t)b * hw + pix) * 2) = g21;
    *reinterpret_cast<float2*>(p.grad_flow12 + ((int64_t)b * hw + pix) * 2) = g12;
    u12 = max(__float_as_uint(g12.x) & 0x7fffffffu, __float_as_uint(g12.y) & 0x7fffffffu);
    u21 = max(__float_as_uint(g21.x) & 0x7fffffffu, __float_as_uint(g21.y) & 0x7fffffffu);
    }();
    // the largest |gradient| per image and direction, for the consumer of these gradients that scales them into fixed
    // point (mr_render_flow_backward's grad_bound: it saves that kernel a pass over its inputs).  Wave maximum at a
    // converged point, then one atomic per wave and direction -- and only while it still raises the stored value
    // (nine tenths of a rendered frame's pixels have a zero gradient).
    if (p.grad_max && __ballot((u12 | u21) != 0u) != 0ull) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            u12 = max(u12, (unsigned)__shfl_xor((int)u12, off));
            u21 = max(u21, (unsigned)__shfl_xor((int)u21, off));
        }
        if ((threadIdx.x & 63) == 0) {
            const unsigned b = lid / (unsigned)p.ntiles;
            if (u12 > p.grad_max[b]) atomicMax(&p.grad_max[b], u12);
            if (u21 > p.grad_max[p.B + b]) atomicMax(&p.grad_max[p.B + b], u21);
        }
    }
    }
}


// ---------------------------------------------------------------------------------------
// Listed launches (mr_*_tiles): the warp half of the training path over the render's tile list
// ---------------------------------------------------------------------------------------
// A frame pair's hand + object meshes cover a sixth of the screen.  The dense kernels above launch a workgroup for
// every block of every image (16 384 of them at B = 64, 256 x 256), five sixths of which read their coverage words and
// leave -- residency x lifetime of those workgroups is half of the dense launch (27 / 23 / 21 us of 47 / 45 / 39), and
// the epilogue and the pair backward still WRITE their zeros under them (98 + 66 MB per step, for no reader: every
// consumer on the training path consults the coverage bytes first).  The render of the stacked pair (2B images: frame
// 1 of every pair, then frame 2) already holds the list of the tiles with candidate faces (raster_fwd.hip:
// bin_boxes_kernel); everything the warp half computes about direction 1 -> 2 lives at pixels frame 1's render covers
// (occl1, flow12, the pair loss's backward term, grad_flow12), and likewise for 2 -> 1 -- so ONE list entry (image of
// the stack, 32 x 8 raster tile) is one workgroup's work in all three kernels, one pixel per thread, and nothing is
// dispatched, read or written for the rest of the screen.  Granularity of the contract: a tile whose 4-byte coverage
// word is non-zero gets all its pixels (inside the crop) written; tiles with a zero word -- listed or not -- get
// NOTHING, and no reader may use what it finds there (round 6: the fused forward reads a value next to the byte that guards it
// and drops it by a select -- addresses inside the planes, contents never used).  Same XCD slices as the render's tile kernel (list_slice): what it wrote for
// a tile is read back behind the same L2.
struct OcclTilesParams {
    const float* mask[2];    // mask_flow1 / mask_flow2  [B,is,is]
    const float* flow[2];    // flow12 / flow21 planes (batch stride fbstride)
    const float* scale[2];   // nullable
    float* occl[2];
    float* out[2];           // [B,crop_h,crop_w,2]
    const uint8_t* hit[2];
    int64_t fbstride;
    int B, is, crop_h, crop_w, tiles_x, tiles_y;
    float dist_thresh, wthresh;
    ListArgs list;
};

__global__ void __launch_bounds__(256) occlusion_flow_tiles_kernel(OcclTilesParams p) {
    unsigned j;
    const ListSlice sl = list_slice(p.list.tlist->n_heavy, p.list.tlist->n_light, j);
    const int T = p.tiles_x * p.tiles_y, is = p.is;
    const int64_t hw = (int64_t)is * is;
    for (; j < sl.n_local; j += sl.stride) {
        const TileAt t = tile_at(p.list.ids[sl.slot(j, p.list.cap)].x, p.B, p.tiles_x, T, is, p.hit[0], p.hit[1]);
        if (t.word == 0u || t.x >= is || t.ry >= is) continue;  // (uniform | border tiles of rasters that are no multiple of the tile)
        const int a = t.dir, o_ = 1 - t.dir;
        const float* ma = p.mask[a] + (int64_t)t.b * hw;
        const float* mb = p.mask[o_] + (int64_t)t.b * hw;
        const float* fab = p.flow[a] + (int64_t)t.b * p.fbstride;
        const float* fba = p.flow[o_] + (int64_t)t.b * p.fbstride;
        const float* sab = p.scale[a] ? p.scale[a] + (int64_t)t.b * hw : nullptr;
        const float* sba = p.scale[o_] ? p.scale[o_] + (int64_t)t.b * hw : nullptr;
        const uint8_t* ha = p.hit[a] + (int64_t)t.b * T * 4;
        const uint8_t* hb = p.hit[o_] + (int64_t)t.b * T * 4;
        const int64_t pix = (int64_t)t.y * is + t.x;
        float o = 0.0f;
        if (t.row_covered) o = occl_one(ma, mb, fab, fba, sab, sba, hw, is, is, t.x, t.y, p.dist_thresh, p.wthresh, ha, hb, p.tiles_x);
        p.occl[a][(int64_t)t.b * hw + pix] = o;
        if (t.y < p.crop_h && t.x < p.crop_w) {
            // (an occlusion bit can only be non-zero inside its own image's coverage: occl_one's guarded reads came first)
            const float sc = (o != 0.0f && sab) ? sab[pix] : 1.0f;
            const float post = o != 0.0f ? ma[pix] * o : 0.0f;
            float2 r = make_float2(0.0f, 0.0f);
            if (post != 0.0f && sc != 0.0f) r = make_float2((fab[pix] * sc) * post, (fab[hw + pix] * sc) * post);
            *reinterpret_cast<float2*>(p.out[a] + (((int64_t)t.b * p.crop_h + t.y) * p.crop_w + t.x) * 2) = r;
        }
    }
}

struct PairTilesParams {
    const float* flow[2];     // flow12 / flow21 [B,H,W,2]
    const float* image_ref;   // [B,3,H,W]
    const float* image;
    const float* jitter_ref;  // [B,Cj,H,W]
    const float* jitter;
    int Cj;
    float* partial;           // forward: [2B, T, 2] {masked L1 sum, 3 x valid count} of stack image i's direction
    const uint8_t* hit[2];
    int B, H, W, is, tiles_x, tiles_y;
    float thresh;
    ListArgs list;
    // backward
    const float* sums;        // [B,4]
    const float* grad_loss_fwd;
    const float* grad_loss_bwd;  // nullable
    float* grad_flow[2];
    unsigned* grad_max;       // nullable, [2B]
};

// direction 0 = frame 1's pixel grid: flow12 warps `image` towards image_ref, gated by jitter_ref (imgflowarp.py:84,82,88,
// 99-107: the loss's "backward" term, sums[2..3]); direction 1 = frame 2's grid: flow21 warps image_ref towards image,
// gated by jitter (:80,85-87,93-102: the "forward" term, sums[0..1])
struct PairDir {
    const float* src;
    const float* tgt;
    const float* jit;
};
__device__ __forceinline__ PairDir pair_dir(const PairTilesParams& p, int dir) {
    PairDir d;
    d.src = dir ? p.image_ref : p.image;
    d.tgt = dir ? p.image : p.image_ref;
    d.jit = dir ? p.jitter : p.jitter_ref;
    return d;
}

__global__ void __launch_bounds__(256, 6) pair_consist_forward_tiles_kernel(PairTilesParams p) {
    __shared__ float red[2][4][2];
    unsigned j;
    const ListSlice sl = list_slice(p.list.tlist->n_heavy, p.list.tlist->n_light, j);
    const int T = p.tiles_x * p.tiles_y;
    const int64_t hw = (int64_t)p.H * p.W;
    unsigned round = 0;
    for (; j < sl.n_local; j += sl.stride) {
        const TileAt t = tile_at(p.list.ids[sl.slot(j, p.list.cap)].x, p.B, p.tiles_x, T, p.is, p.hit[0], p.hit[1]);
        if (t.word == 0u) continue;
        float sum = 0.0f, cnt = 0.0f;
        if (t.row_covered && t.x < p.W && t.y >= 0 && t.y < p.H) {
            const int64_t pix = (int64_t)t.y * p.W + t.x;
            const float2 uv = pair_flow(p.flow[t.dir], t.b, t.x, t.y, p.H, p.W);
            // a pixel whose flow has a zero x component is invalid whatever the images hold (imgflowarp.py:93-100)
            if (uv.x != 0.0f) {
                const PairDir d = pair_dir(p, t.dir);
                const DirTaps tp = pair_taps(uv, t.x, t.y, p.H, p.W);
                DirRaw2 q{};
                pair_load(tp, d.src, d.tgt, d.jit, d.jit, p.Cj, false, t.b, pix, hw, q);
                pin(q);
                const DirRaw r = unpack(q, tp.a);
                const DirOut e = pair_eval(tp, r, p.H, p.W, p.thresh, false);
                if (e.valid) {
#pragma unroll
                    for (int c = 0; c < 3; c++) sum += fabsf(e.s[c] - r.tgt[c]);
                    cnt = 3.0f;
                }
            }
        }
        // block sums, fixed order (wave butterflies, then the four wave partials in wave order); the LDS slots alternate
        // between rounds so that a round needs ONE barrier
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { sum += __shfl_xor(sum, off); cnt += __shfl_xor(cnt, off); }
        // (`round` counts EXECUTED rounds -- tiles with a zero coverage word `continue` above: two consecutive executed rounds
        // must not share a slot, or a wave of the next round could overwrite it while thread 0 still reads)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = round++ & 1u;
        if (lane == 0) { red[slot][wave][0] = sum; red[slot][wave][1] = cnt; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float* o = p.partial + ((int64_t)t.img * T + t.tile) * 2;
            o[0] = red[slot][0][0] + red[slot][1][0] + red[slot][2][0] + red[slot][3][0];
            o[1] = red[slot][0][1] + red[slot][1][1] + red[slot][2][1] + red[slot][3][1];
        }
    }
}

// occlusion_flow_tiles_kernel + pair_consist_forward_tiles_kernel in ONE pass (mr_flow_pair_forward_tiles): the thread that
// has just formed its pixel's final flow is the one that warps the image with it.  Same arithmetic, same per-tile partial
// sums (bit-identical losses); what disappears is the second kernel's chain of dependent loads (list counters -> entry ->
// coverage word -> flow) in front of its taps, and a launch.  The pixel's own image values (target, direct jitter) do
// not depend on the flow: they are requested before the occlusion check's gathers and have arrived when the taps go out.
struct FlowPairFwdParams {
    OcclTilesParams o;
    const float* image_ref;   // [B,3,H,W], H = o.crop_h, W = o.crop_w
    const float* image;
    const float* jitter_ref;  // [B,Cj,H,W]
    const float* jitter;
    int Cj;
    float* partial;           // [2B, T, 2]
    float thresh;
    // GRAD (mr_flow_pair_forward_grad_tiles): the thread holds everything the pair loss's backward needs -- taps, masks, the
    // epilogue's factors -- so it leaves d(sum of |residuals|) / d(rendered displacement) of its pixel, the "unit gradient"
    // (the backward multiplies by grad_loss / count, one scalar per image), and the tile's largest magnitude
    float* unit_grad;         // [2B, H, W, 2] (written under the covered tiles)
    unsigned* tile_max;       // [2B, T] float bits
    // REC (mr_pair_step_forward, round 6): the render left ONE 16-byte record per pixel {displacement x, y, alpha, mask} instead
    // of the planes of o.mask / o.flow / o.scale (unused then; o.occl may be NULL: nobody reads a pair step's occlusion maps)
    const float4* rec;        // [2B, is, is] image orientation: frame 1's B images, then frame 2's
};

template <bool GRAD, bool REC = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) flow_pair_forward_tiles_kernel(FlowPairFwdParams q) {
    __shared__ float red[2][4][2];
    __shared__ unsigned redm[2][4];
    const OcclTilesParams& p = q.o;
    unsigned j;
    const ListSlice sl = list_slice(p.list.tlist->n_heavy, p.list.tlist->n_light, j);
    const int T = p.tiles_x * p.tiles_y, is = p.is, H = p.crop_h, W = p.crop_w;
    const int64_t hw = (int64_t)is * is, hw_img = (int64_t)H * W;
    unsigned round = 0;
    for (; j < sl.n_local; j += sl.stride) {
        const TileAt t = tile_at(p.list.ids[sl.slot(j, p.list.cap)].x, p.B, p.tiles_x, T, is, p.hit[0], p.hit[1]);
        // Round 6: the pixel's own values are requested TOGETHER with the tile's coverage word, not behind it (nine dependent
        // round trips per tile -- list header, entry, word, own values, byte at q, planes at q, byte at r, mask at r, taps --
        // were this kernel: six now, occl_from_own_together takes two more out).  What the planes hold under a row pair
        // without a covered pixel is discarded below; the addresses are inside the planes whatever the word says.
        const bool inside = t.x < is && t.ry < is;
        const int a = t.dir, o_ = 1 - t.dir;
        const float* ma = REC ? nullptr : p.mask[a] + (int64_t)t.b * hw;
        const float* fab = REC ? nullptr : p.flow[a] + (int64_t)t.b * p.fbstride;
        const float* sab = (!REC && p.scale[a]) ? p.scale[a] + (int64_t)t.b * hw : nullptr;
        const float4* rec_a = REC ? q.rec + ((int64_t)a * p.B + t.b) * hw : nullptr;
        const int64_t pix = (int64_t)t.y * is + t.x;
        const bool in_crop = inside && t.y < H && t.x < W;
        const int64_t pixc = (int64_t)t.y * W + t.x;
        // direction 0 = frame 1's grid: flow12 warps `image` towards image_ref, gated by jitter_ref; direction 1: the reverse
        const float* src = t.dir ? q.image_ref : q.image;
        const float* tgt = t.dir ? q.image : q.image_ref;
        const float* jit = t.dir ? q.jitter : q.jitter_ref;
        DirRaw2 raw{};
        float ma_p = 0.0f, sc = 1.0f, f0 = 0.0f, f1 = 0.0f;
        if (in_crop) pair_load_own(tgt, jit, q.Cj, t.b, pixc, hw_img, raw);
        if (inside) {
            if constexpr (REC) {
                const float4 own = rec_a[pix];
                f0 = own.x; f1 = own.y; sc = own.w; ma_p = record_mask(own, a);
            } else {
                ma_p = ma[pix];
                if (sab) sc = sab[pix];
                f0 = fab[pix]; f1 = fab[hw + pix];
            }
        }
        pin(ma_p); pin(sc); pin(f0); pin(f1);  // (requested in front of the branch on the word, not sunk behind it)
        if (t.word == 0u) continue;  // (uniform)
        float sum = 0.0f, cnt = 0.0f;
        unsigned gmx = 0u;
        if (inside) {
            const uint8_t* ha = p.hit[a] + (int64_t)t.b * T * 4;
            const uint8_t* hb = p.hit[o_] + (int64_t)t.b * T * 4;
            // (the planes are defined wherever the row pair holds a covered pixel: elsewhere the values above are dropped)
            if (!t.row_covered) { ma_p = 0.0f; sc = 1.0f; f0 = 0.0f; f1 = 0.0f; }
            float o = 0.0f;
            if constexpr (REC) {
                if (t.row_covered && ma_p != 0.0f)
                    o = occl_from_own_records(ma_p, f0 * sc, f1 * sc, rec_a, q.rec + ((int64_t)o_ * p.B + t.b) * hw, a, is, is, t.x, t.y,
                                              p.dist_thresh, p.wthresh, ha, hb, p.tiles_x);
            } else {
                const float* mb = p.mask[o_] + (int64_t)t.b * hw;
                const float* fba = p.flow[o_] + (int64_t)t.b * p.fbstride;
                const float* sba = p.scale[o_] ? p.scale[o_] + (int64_t)t.b * hw : nullptr;
                if (t.row_covered && ma_p != 0.0f)
                    o = occl_from_own_together(ma_p, sab ? f0 * sc : f0, sab ? f1 * sc : f1, ma, mb, fba, sba, hw, is, is, t.x, t.y,
                                               p.dist_thresh, p.wthresh, ha, hb, p.tiles_x);
            }
            if (p.occl[a]) p.occl[a][(int64_t)t.b * hw + pix] = o;
            if (in_crop) {
                const float post = o != 0.0f ? ma_p * o : 0.0f;
                float2 r = make_float2(0.0f, 0.0f);
                if (post != 0.0f && sc != 0.0f) r = make_float2((f0 * sc) * post, (f1 * sc) * post);
                *reinterpret_cast<float2*>(p.out[a] + ((int64_t)t.b * hw_img + pixc) * 2) = r;
#pragma unroll
                for (int c = 0; c < 3; c++) pin(raw.tgt[c]);
                pin(raw.jd);
                float2 gq = make_float2(0.0f, 0.0f);
                if (r.x != 0.0f) {  // (a zero x component: invalid whatever the images hold, imgflowarp.py:93-100)
                    const DirTaps tp = pair_taps(r, t.x, t.y, H, W);
                    pair_load_taps(tp, src, jit, q.Cj, t.b, hw_img, raw);
                    pin(raw);
                    const DirRaw rr = unpack(raw, tp.a);
                    const DirOut e = pair_eval(tp, rr, H, W, q.thresh, false);
                    if (e.valid) {
#pragma unroll
                        for (int c = 0; c < 3; c++) sum += fabsf(e.s[c] - rr.tgt[c]);
                        cnt = 3.0f;
                    }
                    if (GRAD) {
                        // pair_consist_backward_tiles_kernel's gradient with coefficient 1, times the epilogue's factors as
                        // the raster backward applies them: (g * (mask_x * occl)) * mask_pre
                        const float2 gp = pair_grad(tp, rr, e, H, W, 1.0f);
                        gq = make_float2((gp.x * post) * sc, (gp.y * post) * sc);
                    }
                }
                if (GRAD) {
                    *reinterpret_cast<float2*>(q.unit_grad + ((int64_t)t.img * hw_img + pixc) * 2) = gq;
                    gmx = max(__float_as_uint(gq.x) & 0x7fffffffu, __float_as_uint(gq.y) & 0x7fffffffu);
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { sum += __shfl_xor(sum, off); cnt += __shfl_xor(cnt, off); }
        if (GRAD) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) gmx = max(gmx, (unsigned)__shfl_xor((int)gmx, off));
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = round++ & 1u;  // (executed rounds only: see above)
        if (lane == 0) { red[slot][wave][0] = sum; red[slot][wave][1] = cnt; if (GRAD) redm[slot][wave] = gmx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float* out = q.partial + ((int64_t)t.img * T + t.tile) * 2;
            out[0] = red[slot][0][0] + red[slot][1][0] + red[slot][2][0] + red[slot][3][0];
            out[1] = red[slot][0][1] + red[slot][1][1] + red[slot][2][1] + red[slot][3][1];
            if (GRAD) q.tile_max[(int64_t)t.img * T + t.tile] = max(max(redm[slot][0], redm[slot][1]), max(redm[slot][2], redm[slot][3]));
        }
    }
}

// per-sample fixed-order reduction of the tile partials, counted only where the coverage words say a workgroup wrote one
// (the words and the partials of a tile are requested together: whatever an unwritten partial holds is discarded)
__global__ void __launch_bounds__(256) pair_consist_finalize_tiles_kernel(const float* __restrict__ partial,
                                                                          const uint32_t* __restrict__ hit12,
                                                                          const uint32_t* __restrict__ hit21, int B, int T,
                                                                          float* __restrict__ sums, float* __restrict__ loss_fwd,
                                                                          float* __restrict__ loss_bwd,
                                                                          const unsigned* __restrict__ tile_max,
                                                                          unsigned* __restrict__ image_max,
                                                                          float* __restrict__ loss_sum, ScatterWork work,
                                                                          unsigned* __restrict__ mean_ctr,
                                                                          float* __restrict__ mean_out, int mean_of,
                                                                          unsigned* __restrict__ list_reset) {
    // list_reset (mr_pair_step_forward): the three counters of the render's tile-list header.  This launch is the last of the
    // pair's forward side, and nothing behind the fused warp forward reads them: zeroed here, the NEXT pair step on this
    // scratch finds its list header clean (MR_PAIR_STEP_LIST_CLEAN) without a clearing launch in front of its binning pass.
    if (list_reset && blockIdx.x == 0 && threadIdx.x < 3) list_reset[threadIdx.x] = 0u;
    __shared__ float red[4][4];
    __shared__ unsigned redm[4][2];
    __shared__ int wcov[2][4][4];  // [direction][round of the chunk][wave]
    const int b = blockIdx.x;
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    unsigned m12 = 0u, m21 = 0u;  // largest unit-gradient magnitude (float bits) of stack images b and B + b
    if (work.cov) {
        // ... and the ids of the covered tiles of stack images b and B + b, compacted (ascending) for the raster backward's
        // launch (ScatterWork, mr_common.hpp): this workgroup holds the coverage words of both anyway
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int base12 = 0, base21 = 0;
        // FIN_CH rounds of 256 tiles at a time: their words / partials are requested together and ONE barrier pair serves them
        // (round 6; a round per barrier pair before: 4 rounds of dependent loads and 8 barriers at 480 x 480, 7.0 us for the 8
        // workgroups of config 3).  The order of the sums (round by round per thread) and of the lists (ascending) is unchanged.
        constexpr int FIN_CH = 4;
        for (int t0 = 0; t0 < T; t0 += 256 * FIN_CH) {
            uint32_t h21[FIN_CH], h12[FIN_CH];
            float2 v1[FIN_CH], v2[FIN_CH];
            unsigned t21[FIN_CH], t12[FIN_CH];
#pragma unroll
            for (int u = 0; u < FIN_CH; u++) {
                h21[u] = 0u; h12[u] = 0u; v1[u] = make_float2(0.0f, 0.0f); v2[u] = v1[u]; t21[u] = 0u; t12[u] = 0u;
                if (t0 + u * 256 >= T) continue;  // (uniform: a 256-tile raster is ONE round, as before)
                const int t = t0 + u * 256 + (int)threadIdx.x;
                const int tc = t < T ? t : T - 1;
                h21[u] = hit21[(int64_t)b * T + tc]; h12[u] = hit12[(int64_t)b * T + tc];
                v1[u] = *reinterpret_cast<const float2*>(partial + ((int64_t)(B + b) * T + tc) * 2);
                v2[u] = *reinterpret_cast<const float2*>(partial + ((int64_t)b * T + tc) * 2);
                if (tile_max) { t21[u] = tile_max[(int64_t)(B + b) * T + tc]; t12[u] = tile_max[(int64_t)b * T + tc]; }
            }
            bool c21[FIN_CH], c12[FIN_CH];
            unsigned long long k12[FIN_CH], k21[FIN_CH];
#pragma unroll
            for (int u = 0; u < FIN_CH; u++) {
                c21[u] = false; c12[u] = false; k12[u] = 0ull; k21[u] = 0ull;
                if (t0 + u * 256 >= T) continue;
                const bool in = t0 + u * 256 + (int)threadIdx.x < T;
                c21[u] = in && h21[u] != 0u; c12[u] = in && h12[u] != 0u;
                if (c21[u]) { a[0] += v1[u].x; a[1] += v1[u].y; m21 = max(m21, t21[u]); }
                if (c12[u]) { a[2] += v2[u].x; a[3] += v2[u].y; m12 = max(m12, t12[u]); }
                k12[u] = __ballot(c12[u]); k21[u] = __ballot(c21[u]);
                if (lane == 0) { wcov[0][u][wave] = __popcll(k12[u]); wcov[1][u][wave] = __popcll(k21[u]); }
            }
            __syncthreads();
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int u = 0; u < FIN_CH; u++) {
                if (t0 + u * 256 >= T) continue;
                int o12 = base12, o21 = base21;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    o12 += w < wave ? wcov[0][u][w] : 0; o21 += w < wave ? wcov[1][u][w] : 0;
                    base12 += wcov[0][u][w]; base21 += wcov[1][u][w];
                }
                const int t = t0 + u * 256 + (int)threadIdx.x;
                if (c12[u]) work.cov[(int64_t)b * T + o12 + __popcll(k12[u] & below)] = (unsigned short)t;
                if (c21[u]) work.cov[(int64_t)(B + b) * T + o21 + __popcll(k21[u] & below)] = (unsigned short)t;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) { work.n_cov[b] = base12; work.n_cov[B + b] = base21; }
    } else {
        for (int t = threadIdx.x; t < T; t += 256) {
            const uint32_t h21 = hit21[(int64_t)b * T + t], h12 = hit12[(int64_t)b * T + t];
            const float2 v1 = *reinterpret_cast<const float2*>(partial + ((int64_t)(B + b) * T + t) * 2);
            const float2 v2 = *reinterpret_cast<const float2*>(partial + ((int64_t)b * T + t) * 2);
            unsigned t21 = 0u, t12 = 0u;
            if (tile_max) { t21 = tile_max[(int64_t)(B + b) * T + t]; t12 = tile_max[(int64_t)b * T + t]; }
            if (h21 != 0u) { a[0] += v1.x; a[1] += v1.y; m21 = max(m21, t21); }
            if (h12 != 0u) { a[2] += v2.x; a[3] += v2.y; m12 = max(m12, t12); }
        }
    }
    if (image_max) {  // (uniform)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            m12 = max(m12, (unsigned)__shfl_xor((int)m12, off));
            m21 = max(m21, (unsigned)__shfl_xor((int)m21, off));
        }
        if ((threadIdx.x & 63) == 0) { redm[threadIdx.x >> 6][0] = m12; redm[threadIdx.x >> 6][1] = m21; }
    }
    block_sum4(a, red);  // (its barrier also publishes redm)
    if (threadIdx.x == 0 && image_max) {
        image_max[b] = max(max(redm[0][0], redm[1][0]), max(redm[2][0], redm[3][0]));
        image_max[B + b] = max(max(redm[0][1], redm[1][1]), max(redm[2][1], redm[3][1]));
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) sums[b * 4 + k] = a[k];
        const float n1 = (a[1] == 0.0f) ? 1.0f : a[1], n2 = (a[3] == 0.0f) ? 1.0f : a[3];
        if (loss_fwd) loss_fwd[b] = a[0] / n1;
        if (loss_bwd) loss_bwd[b] = a[2] / n2;
        if (loss_sum) loss_sum[b] = a[2] / n2 + a[0] / n1;  // pair_consist's warp_loss with use_backward (imgflowarp.py:104-113)
    }
    if (mean_out) {  // (uniform)
        // mr_pair_step_forward: the mean over the batch (warpbranch.py:87-88) by the workgroup that finishes LAST -- no launch of
        // its own (a 64-lane kernel costs 4.8 us on the timeline).  A sample's value crosses workgroups in an agent-scope atomic
        // store, the count in an agent-scope atomic add behind it (s_waitcnt in between: the store has completed), the last
        // arriver reads all B values with agent-scope atomic loads and sums them in a FIXED order (lane-strided partial sums in
        // index order, then a butterfly): the same bits whichever workgroup comes last.  The counter (a spare word of the render's
        // tile-list header, cleared with it) is re-zeroed for the next call.
        __shared__ int s_last_ws;
        if (threadIdx.x == 0) {
            const float n1 = (a[1] == 0.0f) ? 1.0f : a[1], n2 = (a[3] == 0.0f) ? 1.0f : a[3];
            const float mine = mean_of ? a[0] / n1 : a[2] / n2 + a[0] / n1;
            __hip_atomic_store(mean_out + 1 + b, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned old = __hip_atomic_fetch_add(mean_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last_ws = (old == (unsigned)B - 1u) ? 1 : 0;
            if (s_last_ws) __hip_atomic_store(mean_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last_ws && threadIdx.x < 64) {
            float sacc = 0.0f;
            for (int i = threadIdx.x; i < B; i += 64) sacc += __hip_atomic_load(mean_out + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sacc += __shfl_xor(sacc, off);
            if (threadIdx.x == 0) mean_out[0] = sacc / (float)B;
        }
    }
}

__global__ void __launch_bounds__(256) pair_consist_backward_tiles_kernel(PairTilesParams p) {
    unsigned j;
    const ListSlice sl = list_slice(p.list.tlist->n_heavy, p.list.tlist->n_light, j);
    const int T = p.tiles_x * p.tiles_y;
    const int64_t hw = (int64_t)p.H * p.W;
    for (; j < sl.n_local; j += sl.stride) {
        const TileAt t = tile_at(p.list.ids[sl.slot(j, p.list.cap)].x, p.B, p.tiles_x, T, p.is, p.hit[0], p.hit[1]);
        if (t.word == 0u) continue;
        unsigned u = 0u;
        if (t.x < p.W && t.y >= 0 && t.y < p.H) {
            const int64_t pix = (int64_t)t.y * p.W + t.x;
            float2 g = make_float2(0.0f, 0.0f);
            const float c = p.sums[t.b * 4 + (t.dir ? 1 : 3)];
            const float* gl = t.dir ? p.grad_loss_fwd : p.grad_loss_bwd;
            const float coef = gl ? gl[t.b] / ((c == 0.0f) ? 1.0f : c) : 0.0f;
            if (t.row_covered && coef != 0.0f) {
                const float2 uv = pair_flow(p.flow[t.dir], t.b, t.x, t.y, p.H, p.W);
                if (uv.x != 0.0f) {  // the gradient of an invalid pixel is 0 (see the forward kernel)
                    const PairDir d = pair_dir(p, t.dir);
                    const DirTaps tp = pair_taps(uv, t.x, t.y, p.H, p.W);
                    DirRaw2 q{};
                    pair_load(tp, d.src, d.tgt, d.jit, d.jit, p.Cj, false, t.b, pix, hw, q);
                    pin(q);
                    const DirRaw r = unpack(q, tp.a);
                    const DirOut e = pair_eval(tp, r, p.H, p.W, p.thresh, false);
                    g = pair_grad(tp, r, e, p.H, p.W, coef);
                }
            }
            *reinterpret_cast<float2*>(p.grad_flow[t.dir] + ((int64_t)t.b * hw + pix) * 2) = g;
            u = max(__float_as_uint(g.x) & 0x7fffffffu, __float_as_uint(g.y) & 0x7fffffffu);
        }
        // the largest |gradient| per image of the stack (see pair_consist_backward_kernel)
        if (p.grad_max && __ballot(u != 0u) != 0ull) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, off));
            if ((threadIdx.x & 63) == 0 && u > p.grad_max[t.img]) atomicMax(&p.grad_max[t.img], u);
        }
    }
}

}  // namespace mr

using namespace mr;

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

extern "C" int mr_warp_forward(const float* x, const float* flow, float* out, float* mask, int batch_size,
                               int channels, int height, int width, float thresh, int mode,
                               mr_stream_t stream) {
    if (!x || !flow || !out || !mask) return MR_ERR_BADARG;
    if (batch_size < 0 || channels < 0 || height <= 0 || width <= 0 || (mode != 0 && mode != 1))
        return MR_ERR_BADARG;
    const int64_t n = (int64_t)batch_size * height * width;
    if (n == 0 || channels == 0) return MR_OK;
    hipLaunchKernelGGL(warp_forward_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, flow,
                       out, mask, batch_size, channels, height, width, thresh, mode);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_warp_backward(const float* x, const float* flow, const float* grad_out, float* grad_x,
                                float* grad_flow, int batch_size, int channels, int height, int width,
                                float thresh, int mode, mr_stream_t stream) {
    if (!x || !flow || !grad_out) return MR_ERR_BADARG;
    if (batch_size < 0 || channels < 0 || height <= 0 || width <= 0 || (mode != 0 && mode != 1))
        return MR_ERR_BADARG;
    const int64_t n = (int64_t)batch_size * height * width;
    if (n == 0 || (!grad_x && !grad_flow)) return MR_OK;
    hipLaunchKernelGGL(warp_backward_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, flow,
                       grad_out, grad_x, grad_flow, batch_size, channels, height, width, thresh, mode);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_occlusion_mask(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                                 const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                                 const float* flow21_scale, float* occl1, float* occl2, int batch_size,
                                 int height, int width, float distance_thresh, float warp_thresh,
                                 mr_stream_t stream) {
    if (!mask_flow1 || !mask_flow2 || !flow12 || !flow21 || !occl1 || !occl2) return MR_ERR_BADARG;
    if (batch_size < 0 || height <= 0 || width <= 0 || flow_bstride < 2LL * height * width) return MR_ERR_BADARG;
    const int64_t n = (int64_t)batch_size * height * width;
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(occlusion_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, mask_flow1,
                       mask_flow2, flow12, flow21, flow_bstride, flow12_scale, flow21_scale, occl1, occl2, batch_size,
                       height, width, distance_thresh, warp_thresh);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_occlusion_flow(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                                 const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                                 const float* flow21_scale, float* occl1, float* occl2, float* flow_out12,
                                 float* flow_out21, const uint8_t* tile_hit1, const uint8_t* tile_hit2, int batch_size,
                                 int height, int width, int crop_height, int crop_width, float distance_thresh,
                                 float warp_thresh, mr_stream_t stream) {
    if (!mask_flow1 || !mask_flow2 || !flow12 || !flow21 || !occl1 || !occl2 || !flow_out12 || !flow_out21)
        return MR_ERR_BADARG;
    if ((tile_hit1 == nullptr) != (tile_hit2 == nullptr) || (tile_hit1 && height != width)) return MR_ERR_BADARG;
    if (batch_size < 0 || height <= 0 || width <= 0 || flow_bstride < 2LL * height * width) return MR_ERR_BADARG;
    if (crop_height <= 0 || crop_width <= 0 || crop_height > height || crop_width > width) return MR_ERR_BADARG;
    const int64_t n = (int64_t)batch_size * height * width;
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(occlusion_flow_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, mask_flow1,
                       mask_flow2, flow12, flow21, flow_bstride, flow12_scale, flow21_scale, occl1, occl2, flow_out12,
                       flow_out21, batch_size, height, width, crop_height, crop_width, distance_thresh, warp_thresh,
                       tile_hit1, tile_hit2, (width + 31) / 32, (height + 7) / 8);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int64_t mr_pair_consist_workspace_bytes(int batch_size, int height, int width) {
    if (batch_size < 0 || height <= 0 || width <= 0) return MR_ERR_BADARG;
    const int64_t nblk = (int64_t)((width + PT_W - 1) / PT_W) * ((height + PT_H - 1) / PT_H);
    return (int64_t)batch_size * nblk * 4 * (int64_t)sizeof(float);
}

extern "C" int mr_pair_consist_forward(const float* flow12, const float* flow21, const float* image_ref,
                                       const float* image, const float* jitter_ref, const float* jitter,
                                       int jitter_channels, void* workspace, int64_t workspace_bytes,
                                       float* sums, float* loss_fwd, float* loss_bwd, uint8_t* full_mask1,
                                       uint8_t* full_mask2, float* warp_mask1, float* warp_mask2,
                                       float* warp1, float* warp2, float* diff1, float* diff2,
                                       int batch_size, int height, int width, float thresh,
                                       const uint8_t* tile_hit12, const uint8_t* tile_hit21, int hit_image_size,
                                       mr_stream_t stream) {
    if (!flow12 || !flow21 || !image_ref || !image || !jitter_ref || !jitter || !workspace || !sums)
        return MR_ERR_BADARG;
    if ((tile_hit12 || tile_hit21) && (hit_image_size < height || hit_image_size < width)) return MR_ERR_BADARG;
    if (jitter_channels != 1 && jitter_channels != 3) return MR_ERR_BADARG;
    if (batch_size < 0 || height <= 0 || width <= 0) return MR_ERR_BADARG;
    if (workspace_bytes < mr_pair_consist_workspace_bytes(batch_size, height, width)) return MR_ERR_BADARG;
    if (width < 2 || (int64_t)height * width > (1LL << 29)) return MR_ERR_BADARG;  // row-pair taps, 32-bit byte offsets
    if (batch_size == 0) return MR_OK;
    const int tiles_x = (width + PT_W - 1) / PT_W;
    const int nblk = tiles_x * ((height + PT_H - 1) / PT_H);
    if ((int64_t)nblk * batch_size > 0x7fffffffLL) return MR_ERR_BADARG;
    PairParams p{flow12, flow21, image_ref, image, jitter_ref, jitter, jitter_channels, (float*)workspace,
                 full_mask1, full_mask2, warp_mask1, warp_mask2, warp1, warp2, diff1, diff2,
                 batch_size, height, width, nblk, tiles_x, thresh, tile_hit12, tile_hit21, hit_image_size,
                 (hit_image_size + 31) / 32, ((hit_image_size + 31) / 32) * ((hit_image_size + 7) / 8) * 4};
    hipLaunchKernelGGL(pair_consist_forward_kernel, dim3((unsigned)(nblk * batch_size)), dim3(256), 0,
                       (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    hipLaunchKernelGGL(pair_consist_finalize_kernel, dim3(batch_size), dim3(64), 0, (hipStream_t)stream,
                       (const float*)workspace, nblk, sums, loss_fwd, loss_bwd);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_pair_consist_backward(const float* flow12, const float* flow21, const float* image_ref,
                                        const float* image, const float* jitter_ref, const float* jitter,
                                        int jitter_channels, const float* sums, const float* grad_loss_fwd,
                                        const float* grad_loss_bwd, float* grad_flow12, float* grad_flow21,
                                        int batch_size, int height, int width, float thresh,
                                        const uint8_t* tile_hit12, const uint8_t* tile_hit21, int hit_image_size,
                                        float* grad_max, mr_stream_t stream) {
    if (!flow12 || !flow21 || !image_ref || !image || !jitter_ref || !jitter || !sums || !grad_loss_fwd ||
        !grad_flow12 || !grad_flow21)
        return MR_ERR_BADARG;
    if ((tile_hit12 || tile_hit21) && (hit_image_size < height || hit_image_size < width)) return MR_ERR_BADARG;
    if (jitter_channels != 1 && jitter_channels != 3) return MR_ERR_BADARG;
    if (batch_size < 0 || height <= 0 || width <= 0) return MR_ERR_BADARG;
    if (width < 2 || (int64_t)height * width > (1LL << 29)) return MR_ERR_BADARG;  // row-pair taps, 32-bit byte offsets
    if (batch_size == 0) return MR_OK;
    const int tiles_x = (width + PT_W - 1) / PT_W;
    const int nblk = tiles_x * ((height + PT_H - 1) / PT_H);
    if ((int64_t)nblk * batch_size > 0x7fffffffLL) return MR_ERR_BADARG;
    PairBwdParams p{flow12, flow21, image_ref, image, jitter_ref, jitter, jitter_channels, sums,
                    grad_loss_fwd, grad_loss_bwd, grad_flow12, grad_flow21, batch_size, height, width, nblk,
                    tiles_x, thresh, tile_hit12, tile_hit21, hit_image_size, (hit_image_size + 31) / 32,
                    ((hit_image_size + 31) / 32) * ((hit_image_size + 7) / 8) * 4, reinterpret_cast<unsigned*>(grad_max)};
    hipLaunchKernelGGL(pair_consist_backward_kernel, dim3((unsigned)((nblk * batch_size + PT_SUB - 1) / PT_SUB)), dim3(256), 0,
                       (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_abi_version(void) { return MR_ABI_VERSION; }

extern "C" int mr_device_ok(void) {
    // the calling thread's CURRENT device: that is where every entry point of this library launches
    int n = 0, dev = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    const char* a = prop.gcnArchName;
    return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
}

extern "C" int mr_flow_mask(const float* alpha_img, const int32_t* face_index_map, const float* keep_lut, int n_lut,
                            float thresh, float* mask, int batch_size, int image_size, mr_stream_t stream) {
    if (!alpha_img || !mask || batch_size < 0 || image_size <= 0) return MR_ERR_BADARG;
    if (keep_lut && (!face_index_map || n_lut <= 0)) return MR_ERR_BADARG;
    const int64_t n = (int64_t)batch_size * image_size * image_size;
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(flow_mask_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, alpha_img,
                       face_index_map, keep_lut, n_lut, thresh, mask, n, image_size);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_flow_finalize_forward(const float* rgb_img, const float* mask_pre, const float* mask_x,
                                        const float* occl, float* flow, int batch_size, int image_size, int height,
                                        int width, mr_stream_t stream) {
    if (!rgb_img || !mask_pre || !mask_x || !occl || !flow) return MR_ERR_BADARG;
    if (batch_size < 0 || image_size <= 0 || height <= 0 || width <= 0 || height > image_size || width > image_size)
        return MR_ERR_BADARG;
    const int64_t n = (int64_t)batch_size * height * width;
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(flow_finalize_forward_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, rgb_img,
                       mask_pre, mask_x, occl, flow, batch_size, image_size, height, width);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_flow_finalize_backward(const float* grad_flow, const float* mask_pre, const float* mask_x,
                                         const float* occl, float* grad_rgb_img, int batch_size, int image_size,
                                         int height, int width, mr_stream_t stream) {
    if (!grad_flow || !mask_pre || !mask_x || !occl || !grad_rgb_img) return MR_ERR_BADARG;
    if (batch_size < 0 || image_size <= 0 || height <= 0 || width <= 0 || height > image_size || width > image_size)
        return MR_ERR_BADARG;
    const int64_t n = (int64_t)batch_size * image_size * image_size;
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(flow_finalize_backward_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream,
                       grad_flow, mask_pre, mask_x, occl, grad_rgb_img, batch_size, image_size, height, width);
    MR_CHECK_LAUNCH();
    return MR_OK;
}


// ---- listed launches of the warp half (see the kernels) --------------------------------------------------------------
static unsigned listed_grid(int64_t tile_bound, int64_t cap) {
    if (tile_bound <= 0) tile_bound = (cap + 3) / 4;
    const int64_t bound = std::min<int64_t>((cap + 7) & ~(int64_t)7, std::max<int64_t>((tile_bound + 7) & ~(int64_t)7, 8));
    return (unsigned)bound;
}

extern "C" int mr_occlusion_flow_tiles(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                                       const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                                       const float* flow21_scale, float* occl1, float* occl2, float* flow_out12,
                                       float* flow_out21, const uint8_t* tile_hit1, const uint8_t* tile_hit2, int batch_size,
                                       int image_size, int crop_height, int crop_width, float distance_thresh,
                                       float warp_thresh, const void* list_header, const void* list_entries,
                                       int64_t list_capacity, int64_t tile_bound, mr_stream_t stream) {
    if (!mask_flow1 || !mask_flow2 || !flow12 || !flow21 || !occl1 || !occl2 || !flow_out12 || !flow_out21 || !tile_hit1 ||
        !tile_hit2 || !list_header || !list_entries)
        return MR_ERR_BADARG;
    if (batch_size < 0 || image_size <= 0 || flow_bstride < 2LL * image_size * image_size) return MR_ERR_BADARG;
    if (crop_height <= 0 || crop_width <= 0 || crop_height > image_size || crop_width > image_size) return MR_ERR_BADARG;
    const int tiles_x = (image_size + 31) / 32, tiles_y = (image_size + 7) / 8;
    if (list_capacity != 2LL * batch_size * tiles_x * tiles_y || list_capacity > 0x7fffffffLL) return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    OcclTilesParams p{{mask_flow1, mask_flow2}, {flow12, flow21}, {flow12_scale, flow21_scale}, {occl1, occl2},
                      {flow_out12, flow_out21}, {tile_hit1, tile_hit2}, flow_bstride, batch_size, image_size, crop_height,
                      crop_width, tiles_x, tiles_y, distance_thresh, warp_thresh,
                      {(const TileList*)list_header, (const uint4*)list_entries, (unsigned)list_capacity}};
    hipLaunchKernelGGL(occlusion_flow_tiles_kernel, dim3(listed_grid(tile_bound, list_capacity)), dim3(256), 0,
                       (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int64_t mr_pair_consist_tiles_workspace_bytes(int batch_size, int hit_image_size) {
    if (batch_size < 0 || hit_image_size <= 0) return MR_ERR_BADARG;
    // per tile of the 2B stacked images: {sum, count} of the pair loss + (mr_flow_pair_forward_grad_tiles) the largest unit gradient
    return 2LL * batch_size * ((hit_image_size + 31) / 32) * ((hit_image_size + 7) / 8) * 3 * (int64_t)sizeof(float);
}

extern "C" int64_t mr_flow_pair_scatter_work_bytes(int batch_size, int image_size) {
    if (batch_size < 0 || image_size <= 0) return MR_ERR_BADARG;
    return scatter_work_bytes(2 * batch_size, ((image_size + 31) / 32) * ((image_size + 7) / 8));
}

static int pair_tiles_args_ok(const float* flow12, const float* flow21, const float* image_ref, const float* image,
                              const float* jitter_ref, const float* jitter, int jitter_channels, int batch_size, int height,
                              int width, const uint8_t* tile_hit12, const uint8_t* tile_hit21, int hit_image_size,
                              const void* list_header, const void* list_entries, int64_t list_capacity) {
    if (!flow12 || !flow21 || !image_ref || !image || !jitter_ref || !jitter || !tile_hit12 || !tile_hit21 || !list_header ||
        !list_entries)
        return MR_ERR_BADARG;
    if (jitter_channels != 1 && jitter_channels != 3) return MR_ERR_BADARG;
    if (batch_size < 0 || height <= 0 || width <= 0 || hit_image_size < height || hit_image_size < width) return MR_ERR_BADARG;
    if (width < 2 || (int64_t)height * width > (1LL << 29)) return MR_ERR_BADARG;  // row-pair taps, 32-bit byte offsets
    const int64_t T = (int64_t)((hit_image_size + 31) / 32) * ((hit_image_size + 7) / 8);
    if (list_capacity != 2LL * batch_size * T || list_capacity > 0x7fffffffLL) return MR_ERR_BADARG;
    return MR_OK;
}

extern "C" int mr_pair_consist_forward_tiles(const float* flow12, const float* flow21, const float* image_ref,
                                             const float* image, const float* jitter_ref, const float* jitter,
                                             int jitter_channels, void* workspace, int64_t workspace_bytes, float* sums,
                                             float* loss_fwd, float* loss_bwd, int batch_size, int height, int width,
                                             float thresh, const uint8_t* tile_hit12, const uint8_t* tile_hit21,
                                             int hit_image_size, const void* list_header, const void* list_entries,
                                             int64_t list_capacity, int64_t tile_bound, mr_stream_t stream) {
    const int rc = pair_tiles_args_ok(flow12, flow21, image_ref, image, jitter_ref, jitter, jitter_channels, batch_size, height,
                                      width, tile_hit12, tile_hit21, hit_image_size, list_header, list_entries, list_capacity);
    if (rc != MR_OK) return rc;
    if (!workspace || !sums || workspace_bytes < mr_pair_consist_tiles_workspace_bytes(batch_size, hit_image_size))
        return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    PairTilesParams p{};
    p.flow[0] = flow12; p.flow[1] = flow21;
    p.image_ref = image_ref; p.image = image; p.jitter_ref = jitter_ref; p.jitter = jitter; p.Cj = jitter_channels;
    p.partial = (float*)workspace;
    p.hit[0] = tile_hit12; p.hit[1] = tile_hit21;
    p.B = batch_size; p.H = height; p.W = width; p.is = hit_image_size;
    p.tiles_x = (hit_image_size + 31) / 32; p.tiles_y = (hit_image_size + 7) / 8;
    p.thresh = thresh;
    p.list = ListArgs{(const TileList*)list_header, (const uint4*)list_entries, (unsigned)list_capacity};
    hipLaunchKernelGGL(pair_consist_forward_tiles_kernel, dim3(listed_grid(tile_bound, list_capacity)), dim3(256), 0,
                       (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    hipLaunchKernelGGL(pair_consist_finalize_tiles_kernel, dim3(batch_size), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, reinterpret_cast<const uint32_t*>(tile_hit12),
                       reinterpret_cast<const uint32_t*>(tile_hit21), batch_size, p.tiles_x * p.tiles_y, sums, loss_fwd, loss_bwd,
                       (const unsigned*)nullptr, (unsigned*)nullptr, (float*)nullptr, ScatterWork{nullptr, nullptr},
                       (unsigned*)nullptr, (float*)nullptr, 0, (unsigned*)nullptr);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_pair_consist_backward_tiles(const float* flow12, const float* flow21, const float* image_ref,
                                              const float* image, const float* jitter_ref, const float* jitter,
                                              int jitter_channels, const float* sums, const float* grad_loss_fwd,
                                              const float* grad_loss_bwd, float* grad_flow12, float* grad_flow21,
                                              int batch_size, int height, int width, float thresh,
                                              const uint8_t* tile_hit12, const uint8_t* tile_hit21, int hit_image_size,
                                              float* grad_max, const void* list_header, const void* list_entries,
                                              int64_t list_capacity, int64_t tile_bound, mr_stream_t stream) {
    const int rc = pair_tiles_args_ok(flow12, flow21, image_ref, image, jitter_ref, jitter, jitter_channels, batch_size, height,
                                      width, tile_hit12, tile_hit21, hit_image_size, list_header, list_entries, list_capacity);
    if (rc != MR_OK) return rc;
    if (!sums || !grad_loss_fwd || !grad_flow12 || !grad_flow21) return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    PairTilesParams p{};
    p.flow[0] = flow12; p.flow[1] = flow21;
    p.image_ref = image_ref; p.image = image; p.jitter_ref = jitter_ref; p.jitter = jitter; p.Cj = jitter_channels;
    p.hit[0] = tile_hit12; p.hit[1] = tile_hit21;
    p.B = batch_size; p.H = height; p.W = width; p.is = hit_image_size;
    p.tiles_x = (hit_image_size + 31) / 32; p.tiles_y = (hit_image_size + 7) / 8;
    p.thresh = thresh;
    p.list = ListArgs{(const TileList*)list_header, (const uint4*)list_entries, (unsigned)list_capacity};
    p.sums = sums; p.grad_loss_fwd = grad_loss_fwd; p.grad_loss_bwd = grad_loss_bwd;
    p.grad_flow[0] = grad_flow12; p.grad_flow[1] = grad_flow21;
    p.grad_max = reinterpret_cast<unsigned*>(grad_max);
    hipLaunchKernelGGL(pair_consist_backward_tiles_kernel, dim3(listed_grid(tile_bound, list_capacity)), dim3(256), 0,
                       (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}


static int flow_pair_forward_tiles(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                                   const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                                   const float* flow21_scale, float* occl1, float* occl2, float* flow_out12,
                                   float* flow_out21, const uint8_t* tile_hit1, const uint8_t* tile_hit2,
                                   const float* image_ref, const float* image, const float* jitter_ref,
                                   const float* jitter, int jitter_channels, void* workspace, int64_t workspace_bytes,
                                   float* sums, float* loss_fwd, float* loss_bwd, int batch_size, int image_size,
                                   int height, int width, float distance_thresh, float warp_thresh, float pair_thresh,
                                   const void* list_header, const void* list_entries, int64_t list_capacity,
                                   int64_t tile_bound, float* unit_grad, float* unit_grad_max, float* loss_sum,
                                   void* scatter_work, mr_stream_t stream, float* mean_out = nullptr, int mean_of = 0,
                                   int reset_list = 0, const void* records = nullptr) {
    // (`records`, mr_pair_step_forward: the render's 16-byte pixel records [2B,is,is] in place of the mask / flow / scale
    // planes, which are not looked at then; occl1 / occl2 may be NULL -- the occlusion maps are not kept)
    if (records ? (!unit_grad || ((uintptr_t)records & 15)) : (!mask_flow1 || !mask_flow2 || !flow12 || !flow21 || !occl1 || !occl2))
        return MR_ERR_BADARG;
    if (!flow_out12 || !flow_out21) return MR_ERR_BADARG;
    if (batch_size < 0 || image_size <= 0 || (!records && flow_bstride < 2LL * image_size * image_size)) return MR_ERR_BADARG;
    const int rc = pair_tiles_args_ok(flow_out12, flow_out21, image_ref, image, jitter_ref, jitter, jitter_channels, batch_size,
                                      height, width, tile_hit1, tile_hit2, image_size, list_header, list_entries, list_capacity);
    if (rc != MR_OK) return rc;
    if (!workspace || !sums || workspace_bytes < mr_pair_consist_tiles_workspace_bytes(batch_size, image_size))
        return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    const int tiles_x = (image_size + 31) / 32, tiles_y = (image_size + 7) / 8;
    FlowPairFwdParams q{};
    q.o = OcclTilesParams{{mask_flow1, mask_flow2}, {flow12, flow21}, {flow12_scale, flow21_scale}, {occl1, occl2},
                          {flow_out12, flow_out21}, {tile_hit1, tile_hit2}, flow_bstride, batch_size, image_size, height,
                          width, tiles_x, tiles_y, distance_thresh, warp_thresh,
                          {(const TileList*)list_header, (const uint4*)list_entries, (unsigned)list_capacity}};
    q.image_ref = image_ref; q.image = image; q.jitter_ref = jitter_ref; q.jitter = jitter; q.Cj = jitter_channels;
    q.partial = (float*)workspace; q.thresh = pair_thresh;
    q.unit_grad = unit_grad;
    q.tile_max = reinterpret_cast<unsigned*>(q.partial + 2LL * batch_size * tiles_x * tiles_y * 2);
    q.rec = (const float4*)records;
    if (records)
        hipLaunchKernelGGL((flow_pair_forward_tiles_kernel<true, true>), dim3(listed_grid(tile_bound, list_capacity)), dim3(256), 0,
                           (hipStream_t)stream, q);
    else if (unit_grad)
        hipLaunchKernelGGL(flow_pair_forward_tiles_kernel<true>, dim3(listed_grid(tile_bound, list_capacity)), dim3(256), 0,
                           (hipStream_t)stream, q);
    else
        hipLaunchKernelGGL(flow_pair_forward_tiles_kernel<false>, dim3(listed_grid(tile_bound, list_capacity)), dim3(256), 0,
                           (hipStream_t)stream, q);
    MR_CHECK_LAUNCH();
    hipLaunchKernelGGL(pair_consist_finalize_tiles_kernel, dim3(batch_size), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, reinterpret_cast<const uint32_t*>(tile_hit1),
                       reinterpret_cast<const uint32_t*>(tile_hit2), batch_size, tiles_x * tiles_y, sums, loss_fwd, loss_bwd,
                       unit_grad ? (const unsigned*)q.tile_max : (const unsigned*)nullptr,
                       unit_grad ? reinterpret_cast<unsigned*>(unit_grad_max) : (unsigned*)nullptr, loss_sum,
                       scatter_work_at(unit_grad ? scatter_work : nullptr, 2 * batch_size),
                       // (the arrival counter of the batch mean: a spare word of the list header the caller cleared with it)
                       mean_out ? const_cast<unsigned*>(&((const TileList*)list_header)->pad[0]) : (unsigned*)nullptr, mean_out, mean_of,
                       reset_list ? const_cast<unsigned*>(&((const TileList*)list_header)->n_heavy) : (unsigned*)nullptr);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

// mr_flow_pair_forward_grad_tiles + the mean over the batch (mr_pair_step_forward, pair_step.hip): mean_out[0] = the mean of
// loss_bwd + loss_fwd (mean_of = 0) or of loss_fwd (1); mean_out[1 .. B] scratch.  The list header's spare words must have
// been cleared with the header (MR_FLAG_TILE_LIST_CLEARED's region: the pair prologue does it).
int mr_flow_pair_forward_grad_tiles_ex(const float* mask_flow1, const float* mask_flow2, const float* flow12, const float* flow21,
                                       int64_t flow_bstride, const float* flow12_scale, const float* flow21_scale, float* occl1,
                                       float* occl2, float* flow_out12, float* flow_out21, const uint8_t* tile_hit1,
                                       const uint8_t* tile_hit2, const float* image_ref, const float* image, const float* jitter_ref,
                                       const float* jitter, int jitter_channels, void* workspace, int64_t workspace_bytes,
                                       float* sums, float* loss_fwd, float* loss_bwd, int batch_size, int image_size, int height,
                                       int width, float distance_thresh, float warp_thresh, float pair_thresh, const void* list_header,
                                       const void* list_entries, int64_t list_capacity, int64_t tile_bound, float* unit_grad,
                                       float* unit_grad_max, float* loss_sum, void* scatter_work, float* mean_out, int mean_of,
                                       int reset_list, const void* records, mr_stream_t stream) {
    if (!unit_grad || !unit_grad_max) return MR_ERR_BADARG;
    return flow_pair_forward_tiles(mask_flow1, mask_flow2, flow12, flow21, flow_bstride, flow12_scale, flow21_scale, occl1, occl2,
                                   flow_out12, flow_out21, tile_hit1, tile_hit2, image_ref, image, jitter_ref, jitter,
                                   jitter_channels, workspace, workspace_bytes, sums, loss_fwd, loss_bwd, batch_size, image_size,
                                   height, width, distance_thresh, warp_thresh, pair_thresh, list_header, list_entries,
                                   list_capacity, tile_bound, unit_grad, unit_grad_max, loss_sum, scatter_work, stream, mean_out,
                                   mean_of, reset_list, records);
}

extern "C" int mr_flow_pair_forward_tiles(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                                          const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                                          const float* flow21_scale, float* occl1, float* occl2, float* flow_out12,
                                          float* flow_out21, const uint8_t* tile_hit1, const uint8_t* tile_hit2,
                                          const float* image_ref, const float* image, const float* jitter_ref,
                                          const float* jitter, int jitter_channels, void* workspace, int64_t workspace_bytes,
                                          float* sums, float* loss_fwd, float* loss_bwd, int batch_size, int image_size,
                                          int height, int width, float distance_thresh, float warp_thresh, float pair_thresh,
                                          const void* list_header, const void* list_entries, int64_t list_capacity,
                                          int64_t tile_bound, mr_stream_t stream) {
    return flow_pair_forward_tiles(mask_flow1, mask_flow2, flow12, flow21, flow_bstride, flow12_scale, flow21_scale, occl1, occl2,
                                   flow_out12, flow_out21, tile_hit1, tile_hit2, image_ref, image, jitter_ref, jitter,
                                   jitter_channels, workspace, workspace_bytes, sums, loss_fwd, loss_bwd, batch_size, image_size,
                                   height, width, distance_thresh, warp_thresh, pair_thresh, list_header, list_entries,
                                   list_capacity, tile_bound, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int mr_flow_pair_forward_grad_tiles(const float* mask_flow1, const float* mask_flow2, const float* flow12,
                                               const float* flow21, int64_t flow_bstride, const float* flow12_scale,
                                               const float* flow21_scale, float* occl1, float* occl2, float* flow_out12,
                                               float* flow_out21, const uint8_t* tile_hit1, const uint8_t* tile_hit2,
                                               const float* image_ref, const float* image, const float* jitter_ref,
                                               const float* jitter, int jitter_channels, void* workspace,
                                               int64_t workspace_bytes, float* sums, float* loss_fwd, float* loss_bwd,
                                               int batch_size, int image_size, int height, int width, float distance_thresh,
                                               float warp_thresh, float pair_thresh, const void* list_header,
                                               const void* list_entries, int64_t list_capacity, int64_t tile_bound,
                                               float* unit_grad, float* unit_grad_max, float* loss_sum,
                                               void* scatter_work, mr_stream_t stream) {
    if (!unit_grad || !unit_grad_max) return MR_ERR_BADARG;
    return flow_pair_forward_tiles(mask_flow1, mask_flow2, flow12, flow21, flow_bstride, flow12_scale, flow21_scale, occl1, occl2,
                                   flow_out12, flow_out21, tile_hit1, tile_hit2, image_ref, image, jitter_ref, jitter,
                                   jitter_channels, workspace, workspace_bytes, sums, loss_fwd, loss_bwd, batch_size, image_size,
                                   height, width, distance_thresh, warp_thresh, pair_thresh, list_header, list_entries,
                                   list_capacity, tile_bound, unit_grad, unit_grad_max, loss_sum, scatter_work, stream);
}
