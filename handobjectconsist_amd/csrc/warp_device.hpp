// warp_device.hpp -- device functions of the photometric-warp kernels that more than one translation unit needs
// (warp.hip: the warp / occlusion / pair-loss kernels; raster_bwd.hip: the fused pair-loss + raster backward of the
// training path).  grid_sample semantics restated (torch, zeros padding, align_corners=False):
//   vx = 2 (x + u) / max(W - 1, 1) - 1 ;  ix = ((vx + 1) W - 1) / 2
#pragma once

#include "mr_common.hpp"

namespace mr {

struct Taps {
    int x0, y0;            // north-west tap
    float nw, ne, sw, se;  // bilinear weights
    float ix, iy;
};

__device__ __forceinline__ void sample_pos(float x, float y, float u, float v, int W, int H, float& ix,
                                           float& iy) {
    const float gx = x + u, gy = y + v;
    const float vx = 2.0f * gx / (float)max(W - 1, 1) - 1.0f;
    const float vy = 2.0f * gy / (float)max(H - 1, 1) - 1.0f;
    ix = ((vx + 1.0f) * (float)W - 1.0f) / 2.0f;
    iy = ((vy + 1.0f) * (float)H - 1.0f) / 2.0f;
}

__device__ __forceinline__ Taps make_taps(float ix, float iy) {
    Taps t;
    t.ix = ix; t.iy = iy;
    const float fx = floorf(ix), fy = floorf(iy);
    // clamp before the int conversion so that huge / non-finite positions are simply out of bounds
    t.x0 = (int)fminf(fmaxf(fx, -4.0f), 1.0e9f);
    t.y0 = (int)fminf(fmaxf(fy, -4.0f), 1.0e9f);
    if (!(fx == fx)) t.x0 = -4;
    if (!(fy == fy)) t.y0 = -4;
    const float ix_se = fx + 1.0f, iy_se = fy + 1.0f;
    t.nw = (ix_se - ix) * (iy_se - iy);
    t.ne = (ix - fx) * (iy_se - iy);
    t.sw = (ix_se - ix) * (iy - fy);
    t.se = (ix - fx) * (iy - fy);
    return t;
}

__device__ __forceinline__ bool inb(int x, int y, int W, int H) { return x >= 0 && x < W && y >= 0 && y < H; }

// bilinear sample of one channel plane (zeros padding), accumulation order nw, ne, sw, se.
// The four loads are unconditional (clamped addresses) so they issue back to back; a tap that
// is out of bounds leaves the accumulator untouched, exactly like the skipped branch of the
// reference implementation.
struct TapAddr {
    int64_t a_nw, a_ne, a_sw, a_se;
    bool b_nw, b_ne, b_sw, b_se;
};

__device__ __forceinline__ TapAddr tap_addr(const Taps& t, int W, int H) {
    TapAddr a;
    a.b_nw = inb(t.x0, t.y0, W, H);
    a.b_ne = inb(t.x0 + 1, t.y0, W, H);
    a.b_sw = inb(t.x0, t.y0 + 1, W, H);
    a.b_se = inb(t.x0 + 1, t.y0 + 1, W, H);
    const int xc0 = min(max(t.x0, 0), W - 1), xc1 = min(max(t.x0 + 1, 0), W - 1);
    const int yc0 = min(max(t.y0, 0), H - 1), yc1 = min(max(t.y0 + 1, 0), H - 1);
    a.a_nw = (int64_t)yc0 * W + xc0;
    a.a_ne = (int64_t)yc0 * W + xc1;
    a.a_sw = (int64_t)yc1 * W + xc0;
    a.a_se = (int64_t)yc1 * W + xc1;
    return a;
}

__device__ __forceinline__ float bilin(const float* __restrict__ plane, const Taps& t, const TapAddr& a) {
    const float v_nw = plane[a.a_nw], v_ne = plane[a.a_ne], v_sw = plane[a.a_sw], v_se = plane[a.a_se];
    float acc = 0.0f;
    acc = a.b_nw ? acc + v_nw * t.nw : acc;
    acc = a.b_ne ? acc + v_ne * t.ne : acc;
    acc = a.b_sw ? acc + v_sw * t.sw : acc;
    acc = a.b_se ? acc + v_se * t.se : acc;
    return acc;
}

__device__ __forceinline__ float bilin(const float* __restrict__ plane, const Taps& t, int W, int H) {
    return bilin(plane, t, tap_addr(t, W, H));
}

// bilinear sample of an all-ones image = sum of the in-bounds weights, binarised as
// imgflowarp.py:52-53 (mask[mask < thresh] = 0; mask[mask > 0] = 1)
__device__ __forceinline__ float valid_mask(const Taps& t, int W, int H, float thresh) {
    float acc = 0.0f;
    if (inb(t.x0, t.y0, W, H)) acc += t.nw;
    if (inb(t.x0 + 1, t.y0, W, H)) acc += t.ne;
    if (inb(t.x0, t.y0 + 1, W, H)) acc += t.sw;
    if (inb(t.x0 + 1, t.y0 + 1, W, H)) acc += t.se;
    if (acc < thresh) acc = 0.0f;
    if (acc > 0.0f) acc = 1.0f;
    return acc;
}

// d(sample)/d(ix), d(sample)/d(iy) of one channel plane
__device__ __forceinline__ void bilin_grad(const float* __restrict__ plane, const Taps& t, const TapAddr& a,
                                           float& gix, float& giy) {
    const float fx = (float)t.x0, fy = (float)t.y0;
    const float ix_se = fx + 1.0f, iy_se = fy + 1.0f;
    const float v_nw = plane[a.a_nw], v_ne = plane[a.a_ne], v_sw = plane[a.a_sw], v_se = plane[a.a_se];
    gix = 0.0f; giy = 0.0f;
    if (a.b_nw) { gix -= v_nw * (iy_se - t.iy); giy -= v_nw * (ix_se - t.ix); }
    if (a.b_ne) { gix += v_ne * (iy_se - t.iy); giy -= v_ne * (t.ix - fx); }
    if (a.b_sw) { gix -= v_sw * (t.iy - fy); giy += v_sw * (ix_se - t.ix); }
    if (a.b_se) { gix += v_se * (t.iy - fy); giy += v_se * (t.ix - fx); }
}

__device__ __forceinline__ void bilin_grad(const float* __restrict__ plane, const Taps& t, int W, int H,
                                           float& gix, float& giy) {
    bilin_grad(plane, t, tap_addr(t, W, H), gix, giy);
}

// Raw values of the four taps of one channel plane.  Loading them into a Quad first and pinning
// them with `pin()` keeps the loads UNCONDITIONAL and back to back: otherwise the compiler sinks
// each load into the branch of its in-bounds select and waits for it there (one exposed HBM
// round trip per tap instead of one per batch).
struct Quad {
    float nw, ne, sw, se;
};
__device__ __forceinline__ Quad load_quad(const float* __restrict__ plane, const TapAddr& a) {
    Quad q;
    q.nw = plane[a.a_nw]; q.ne = plane[a.a_ne]; q.sw = plane[a.a_sw]; q.se = plane[a.a_se];
    return q;
}
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(Quad& q) { pin(q.nw); pin(q.ne); pin(q.sw); pin(q.se); }
template <typename A>
__device__ __forceinline__ float bilin_q(const Quad& q, const Taps& t, const A& a) {
    float acc = 0.0f;
    acc = a.b_nw ? acc + q.nw * t.nw : acc;
    acc = a.b_ne ? acc + q.ne * t.ne : acc;
    acc = a.b_sw ? acc + q.sw * t.sw : acc;
    acc = a.b_se ? acc + q.se * t.se : acc;
    return acc;
}
template <typename A>
__device__ __forceinline__ void bilin_grad_q(const Quad& q, const Taps& t, const A& a, float& gix, float& giy) {
    const float fx = (float)t.x0, fy = (float)t.y0;
    const float ix_se = fx + 1.0f, iy_se = fy + 1.0f;
    gix = 0.0f; giy = 0.0f;
    if (a.b_nw) { gix -= q.nw * (iy_se - t.iy); giy -= q.nw * (ix_se - t.ix); }
    if (a.b_ne) { gix += q.ne * (iy_se - t.iy); giy -= q.ne * (t.ix - fx); }
    if (a.b_sw) { gix -= q.sw * (t.iy - fy); giy += q.sw * (ix_se - t.ix); }
    if (a.b_se) { gix += q.se * (t.iy - fy); giy += q.se * (t.ix - fx); }
}

// Row-pair addressing (W >= 2): the west / east taps of a row are adjacent in memory, so ONE
// 8-byte load per row fetches both -- half the load instructions of four scalar taps, which is
// what bounds the fused pair kernels.  The pair starts at column clamp(x0, 0, W - 2); when that
// differs from x0 the in-bounds tap sits in the other half (x0 == -1: east tap = first element;
// x0 == W - 1: west tap = second element); taps that are out of bounds are never used.
// Offsets are 32-bit BYTE offsets from a wave-uniform plane pointer (global_load saddr form).
struct PairAddr {
    unsigned o_n, o_s;  // byte offsets of the pairs in rows y0 and y0 + 1 (both clamped)
    bool shl, shr;      // pair shifted right of x0 (x0 < 0) / left of x0 (x0 > W - 2)
    bool b_nw, b_ne, b_sw, b_se;
};
__device__ __forceinline__ PairAddr pair_addr(const Taps& t, int W, int H) {
    PairAddr a;
    a.b_nw = inb(t.x0, t.y0, W, H);
    a.b_ne = inb(t.x0 + 1, t.y0, W, H);
    a.b_sw = inb(t.x0, t.y0 + 1, W, H);
    a.b_se = inb(t.x0 + 1, t.y0 + 1, W, H);
    const int xs = min(max(t.x0, 0), W - 2);
    const int yc0 = min(max(t.y0, 0), H - 1), yc1 = min(max(t.y0 + 1, 0), H - 1);
    a.o_n = ((unsigned)yc0 * (unsigned)W + (unsigned)xs) * 4u;
    a.o_s = ((unsigned)yc1 * (unsigned)W + (unsigned)xs) * 4u;
    a.shl = t.x0 < 0;
    a.shr = t.x0 > W - 2;
    return a;
}
struct __attribute__((packed, aligned(4))) F2 {
    float x, y;
};
struct Quad2 {
    F2 n, s;
};
__device__ __forceinline__ Quad2 load_quad2(const float* __restrict__ plane, const PairAddr& a) {
    const char* base = reinterpret_cast<const char*>(plane);
    Quad2 q;
    q.n = *reinterpret_cast<const F2*>(base + a.o_n);
    q.s = *reinterpret_cast<const F2*>(base + a.o_s);
    return q;
}
__device__ __forceinline__ void pin(Quad2& q) {
    asm volatile("" : "+v"(q.n.x)); asm volatile("" : "+v"(q.n.y));
    asm volatile("" : "+v"(q.s.x)); asm volatile("" : "+v"(q.s.y));
}
__device__ __forceinline__ Quad quad_of(const Quad2& q, const PairAddr& a) {
    Quad r;
    r.nw = a.shr ? q.n.y : q.n.x; r.ne = a.shl ? q.n.x : q.n.y;
    r.sw = a.shr ? q.s.y : q.s.x; r.se = a.shl ? q.s.x : q.s.y;
    return r;
}

__device__ __forceinline__ void nearest_idx(float ix, float iy, int& xn, int& yn) {
    const float rx = rintf(ix), ry = rintf(iy);  // round half to even, as nearbyint
    xn = (rx == rx) ? (int)fminf(fmaxf(rx, -4.0f), 1.0e9f) : -4;
    yn = (ry == ry) ? (int)fminf(fmaxf(ry, -4.0f), 1.0e9f) : -4;
}

// ---------------------------------------------------------------------------------------
// forward-backward occlusion check, one direction at one pixel
// ---------------------------------------------------------------------------------------
// One direction: occl_a(p) from (mask_a, mask_b, flow_ab, flow_ba).
//   grid_a = [x/W, y/H, mask_a, mask_a]
//   warp_ab(q)  = nearest(grid_a; q + flow_ba(q)) * mask_b(q)            (imgflowarp.py:132,134)
//   warp_aba(p) = nearest(warp_ab; p + flow_ab(p)) * mask_a(p)           (:136,138)
//   occl_a = occlusion_mask_from_warped_grid(grid_a, warp_aba)           (:145)
// Coverage bytes of mr_render_flow_forward ([tiles_y, tiles_x, 4], RASTER orientation, one byte per 32 x 8 tile and row
// pair; the planes read here are in IMAGE orientation): 0 = nothing covered there -- with MR_FLAG_SPARSE_TILES the
// render did not even write those pixels, so every read of a rendered plane is guarded by this test.
__device__ __forceinline__ bool tile_covered(const uint8_t* __restrict__ hit, int tiles_x, int H, int x, int y) {
    if (!hit) return true;
    const int ry = H - 1 - y;
    return hit[((ry >> 3) * tiles_x + (x >> 5)) * 4 + ((ry & 7) >> 1)] != 0;
}

__device__ __forceinline__ float occl_from_own(float ma_p, float fx, float fy, const float* __restrict__ mask_a,
                                               const float* __restrict__ mask_b, const float* __restrict__ flow_ba,
                                               const float* __restrict__ scale_ba, int64_t hw, int H, int W, int xx, int yy,
                                               float dist_thresh, float wthresh, const uint8_t* __restrict__ hit_a,
                                               const uint8_t* __restrict__ hit_b, int tiles_x);

__device__ __forceinline__ float occl_one(const float* __restrict__ mask_a, const float* __restrict__ mask_b,
                                          const float* __restrict__ flow_ab,
                                          const float* __restrict__ flow_ba, const float* __restrict__ scale_ab,
                                          const float* __restrict__ scale_ba, int64_t hw, int H, int W,
                                          int xx, int yy, float dist_thresh, float wthresh,
                                          const uint8_t* __restrict__ hit_a = nullptr,
                                          const uint8_t* __restrict__ hit_b = nullptr, int tiles_x = 0) {
    const int64_t pix = (int64_t)yy * W + xx;
    if (!tile_covered(hit_a, tiles_x, H, xx, yy)) return 0.0f;
    const float ma_p = mask_a[pix];
    // The result is mask_a(p) * (...) * motion with finite factors (masks are 0 / 1 or a rendered alpha): a pixel
    // outside its own mask -- 90 % of a hand + object frame -- is 0 without any of the dependent gathers below.
    if (ma_p == 0.0f) return 0.0f;
    // second warp: sample warp_ab at p + flow_ab(p)   (flow = raw flow * scale when a scale map is given)
    const float sa = scale_ab ? scale_ab[pix] : 1.0f;
    return occl_from_own(ma_p, scale_ab ? flow_ab[pix] * sa : flow_ab[pix], scale_ab ? flow_ab[hw + pix] * sa : flow_ab[hw + pix],
                         mask_a, mask_b, flow_ba, scale_ba, hw, H, W, xx, yy, dist_thresh, wthresh, hit_a, hit_b, tiles_x);
}

// ... from the pixel's OWN values (mask_a(p) != 0 and the scaled flow_ab(p)) on: callers that can request those values
// without waiting for the mask test (the listed kernels: one dependent round trip less per workgroup)
__device__ __forceinline__ float occl_from_own(float ma_p, float fx, float fy, const float* __restrict__ mask_a,
                                               const float* __restrict__ mask_b, const float* __restrict__ flow_ba,
                                               const float* __restrict__ scale_ba, int64_t hw, int H, int W, int xx, int yy,
                                               float dist_thresh, float wthresh, const uint8_t* __restrict__ hit_a,
                                               const uint8_t* __restrict__ hit_b, int tiles_x) {
    float ix, iy;
    sample_pos((float)xx, (float)yy, fx, fy, W, H, ix, iy);
    int qx, qy;
    nearest_idx(ix, iy, qx, qy);
    float wg[3] = {0.0f, 0.0f, 0.0f};  // channels x, y, mask of warp_ab at q
    float m2 = inb(qx, qy, W, H) ? 1.0f : 0.0f;
    if (m2 < wthresh) m2 = 0.0f;
    if (m2 > 0.0f) {
        const int64_t qpix = (int64_t)qy * W + qx;
        // nothing rendered around q: mask_b(q) = 0 zeroes the warped grid, hence the result
        if (!tile_covered(hit_b, tiles_x, H, qx, qy)) return 0.0f;
        // first warp: sample grid_a at q + flow_ba(q)
        float jx, jy;
        const float sb = scale_ba ? scale_ba[qpix] : 1.0f;
        sample_pos((float)qx, (float)qy, scale_ba ? flow_ba[qpix] * sb : flow_ba[qpix],
                   scale_ba ? flow_ba[hw + qpix] * sb : flow_ba[hw + qpix], W, H, jx, jy);
        int rx, ry;
        nearest_idx(jx, jy, rx, ry);
        float m1 = inb(rx, ry, W, H) ? 1.0f : 0.0f;
        if (m1 < wthresh) m1 = 0.0f;
        if (m1 > 0.0f) {
            const float mb_q = mask_b[qpix];
            const float ma_r = tile_covered(hit_a, tiles_x, H, rx, ry) ? mask_a[(int64_t)ry * W + rx] : 0.0f;
            wg[0] = ((float)rx / (float)W) * m1 * mb_q;
            wg[1] = ((float)ry / (float)H) * m1 * mb_q;
            wg[2] = ma_r * m1 * mb_q;
        }
    }
    float w3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) w3[k] = wg[k] * m2 * ma_p;
    const float g0 = (float)xx / (float)W, g1 = (float)yy / (float)H;
    const float mask = ma_p * w3[2];
    const float dx = (w3[0] - g0) * mask, dy = (w3[1] - g1) * mask;
    const float displ = sqrtf(dx * dx + dy * dy);
    const float motion = (displ < dist_thresh) ? 1.0f : 0.0f;
    return mask * motion;
}

// occl_from_own with every hop's loads requested TOGETHER with the coverage byte that guards them (round 6, the fused warp
// forward): the byte of q's row pair, the other frame's mask, scale and displacement at q go out at once -- then r's byte and
// mask_a(r).  The guarded form above waits for a byte before it asks for the planes behind it: four dependent round trips
// (byte at q -> planes at q -> byte at r -> mask at r) where this one has two.  What an uncovered row pair's planes hold (with
// MR_FLAG_SPARSE_TILES: whatever the buffer held) is read and DISCARDED by a select; every address is inside the planes
// (q and r are in-bounds pixels by the m2 / m1 tests).  Same values, same arithmetic, same result.
__device__ __forceinline__ float occl_from_own_together(float ma_p, float fx, float fy, const float* __restrict__ mask_a,
                                                        const float* __restrict__ mask_b, const float* __restrict__ flow_ba,
                                                        const float* __restrict__ scale_ba, int64_t hw, int H, int W, int xx,
                                                        int yy, float dist_thresh, float wthresh,
                                                        const uint8_t* __restrict__ hit_a, const uint8_t* __restrict__ hit_b,
                                                        int tiles_x) {
    float ix, iy;
    sample_pos((float)xx, (float)yy, fx, fy, W, H, ix, iy);
    int qx, qy;
    nearest_idx(ix, iy, qx, qy);
    float wg[3] = {0.0f, 0.0f, 0.0f};  // channels x, y, mask of warp_ab at q
    float m2 = inb(qx, qy, W, H) ? 1.0f : 0.0f;
    if (m2 < wthresh) m2 = 0.0f;
    if (m2 > 0.0f) {
        const int64_t qpix = (int64_t)qy * W + qx;
        const int qry = H - 1 - qy;
        float cov_q = (float)hit_b[((qry >> 3) * tiles_x + (qx >> 5)) * 4 + ((qry & 7) >> 1)];
        float sb = scale_ba ? scale_ba[qpix] : 1.0f;
        float fbx = flow_ba[qpix], fby = flow_ba[hw + qpix], mb_q = mask_b[qpix];
        pin(cov_q); pin(sb); pin(fbx); pin(fby); pin(mb_q);
        // nothing rendered around q: mask_b(q) = 0 zeroes the warped grid, hence the result
        if (cov_q == 0.0f) return 0.0f;
        // first warp: sample grid_a at q + flow_ba(q)
        float jx, jy;
        sample_pos((float)qx, (float)qy, scale_ba ? fbx * sb : fbx, scale_ba ? fby * sb : fby, W, H, jx, jy);
        int rx, ry;
        nearest_idx(jx, jy, rx, ry);
        float m1 = inb(rx, ry, W, H) ? 1.0f : 0.0f;
        if (m1 < wthresh) m1 = 0.0f;
        if (m1 > 0.0f) {
            const int rry = H - 1 - ry;
            float cov_r = (float)hit_a[((rry >> 3) * tiles_x + (rx >> 5)) * 4 + ((rry & 7) >> 1)];
            float ma_raw = mask_a[(int64_t)ry * W + rx];
            pin(cov_r); pin(ma_raw);
            const float ma_r = cov_r != 0.0f ? ma_raw : 0.0f;
            wg[0] = ((float)rx / (float)W) * m1 * mb_q;
            wg[1] = ((float)ry / (float)H) * m1 * mb_q;
            wg[2] = ma_r * m1 * mb_q;
        }
    }
    float w3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) w3[k] = wg[k] * m2 * ma_p;
    const float g0 = (float)xx / (float)W, g1 = (float)yy / (float)H;
    const float mask = ma_p * w3[2];
    const float dx = (w3[0] - g0) * mask, dy = (w3[1] - g1) * mask;
    const float displ = sqrtf(dx * dx + dy * dy);
    const float motion = (displ < dist_thresh) ? 1.0f : 0.0f;
    return mask * motion;
}

// ... and on the 16-byte pixel records of a pair step's render (round 6; raster_fwd.hip FwdParams::rec4: {displacement x,
// displacement y, alpha, mask} per pixel, image orientation) instead of four planes per frame: what the forward-backward check
// reads at q is ONE record of the other frame, at r one word of this frame's.  mr_pair_step_forward's assignment of planes
// (pair_step.hip) is fixed here: frame 1 is masked by its mask plane (record .w), frame 2 by its alpha plane (.z); both
// displacements are scaled by their frame's mask plane (.w).  Arithmetic and order of operations: occl_from_own's.
__device__ __forceinline__ float record_mask(const float4& r, int frame) { return frame == 0 ? r.w : r.z; }
__device__ __forceinline__ float occl_from_own_records(float ma_p, float fx, float fy, const float4* __restrict__ rec_a,
                                                       const float4* __restrict__ rec_b, int frame_a, int H, int W, int xx, int yy,
                                                       float dist_thresh, float wthresh, const uint8_t* __restrict__ hit_a,
                                                       const uint8_t* __restrict__ hit_b, int tiles_x) {
    float ix, iy;
    sample_pos((float)xx, (float)yy, fx, fy, W, H, ix, iy);
    int qx, qy;
    nearest_idx(ix, iy, qx, qy);
    float wg[3] = {0.0f, 0.0f, 0.0f};  // channels x, y, mask of warp_ab at q
    float m2 = inb(qx, qy, W, H) ? 1.0f : 0.0f;
    if (m2 < wthresh) m2 = 0.0f;
    if (m2 > 0.0f) {
        const int64_t qpix = (int64_t)qy * W + qx;
        const int qry = H - 1 - qy;
        float cov_q = (float)hit_b[((qry >> 3) * tiles_x + (qx >> 5)) * 4 + ((qry & 7) >> 1)];
        float4 rq = rec_b[qpix];
        pin(cov_q); pin(rq.x); pin(rq.y); pin(rq.z); pin(rq.w);
        // nothing rendered around q: mask_b(q) = 0 zeroes the warped grid, hence the result
        if (cov_q == 0.0f) return 0.0f;
        const float sb = rq.w, mb_q = record_mask(rq, 1 - frame_a);
        // first warp: sample grid_a at q + flow_ba(q)
        float jx, jy;
        sample_pos((float)qx, (float)qy, rq.x * sb, rq.y * sb, W, H, jx, jy);
        int rx, ry;
        nearest_idx(jx, jy, rx, ry);
        float m1 = inb(rx, ry, W, H) ? 1.0f : 0.0f;
        if (m1 < wthresh) m1 = 0.0f;
        if (m1 > 0.0f) {
            const int rry = H - 1 - ry;
            float cov_r = (float)hit_a[((rry >> 3) * tiles_x + (rx >> 5)) * 4 + ((rry & 7) >> 1)];
            float ma_raw = reinterpret_cast<const float*>(rec_a + ((int64_t)ry * W + rx))[frame_a == 0 ? 3 : 2];
            pin(cov_r); pin(ma_raw);
            const float ma_r = cov_r != 0.0f ? ma_raw : 0.0f;
            wg[0] = ((float)rx / (float)W) * m1 * mb_q;
            wg[1] = ((float)ry / (float)H) * m1 * mb_q;
            wg[2] = ma_r * m1 * mb_q;
        }
    }
    float w3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) w3[k] = wg[k] * m2 * ma_p;
    const float g0 = (float)xx / (float)W, g1 = (float)yy / (float)H;
    const float mask = ma_p * w3[2];
    const float dx = (w3[0] - g0) * mask, dy = (w3[1] - g1) * mask;
    const float displ = sqrtf(dx * dx + dy * dy);
    const float motion = (displ < dist_thresh) ? 1.0f : 0.0f;
    return mask * motion;
}

// ---------------------------------------------------------------------------------------
// pair loss, one direction at one pixel
// ---------------------------------------------------------------------------------------
// one direction at one pixel: warp `src` with `flow`, gate with the jitter mask `jwarp`
// warped by the same flow and with `jdirect` at the pixel, compare with `tgt`.
struct DirOut {
    float s[3];      // warped source * warp mask
    float m;         // warp mask (before the jitter gate)
    float wm[3];     // warp mask after the jitter gate, per jitter channel
    bool valid;
};

// One direction at one pixel, in three steps so that every global load of the pixel is in
// flight before anything waits:  pair_taps (flow -> tap addresses),  pair_load (raw tap values of
// the 3 source channels and of the jitter mask, the target pixel, the direct jitter value),
// pair_eval (masks, warped values).
struct DirTaps {
    Taps t;
    PairAddr a;
    float2 uv;
};
struct DirRaw2 {  // as loaded: one 8-byte pair per tap row
    Quad2 src[3];
    Quad2 jit[3];
    float tgt[3];
    float jd;
};
struct DirRaw {
    Quad src[3];
    Quad jit[3];
    float tgt[3];
    float jd;
};

__device__ __forceinline__ float2 pair_flow(const float* __restrict__ flow, int b, int xx, int yy, int H, int W) {
    return *reinterpret_cast<const float2*>(flow + ((int64_t)b * H * W + (int64_t)yy * W + xx) * 2);
}
// ... guarded by the coverage bytes of the flow's render (hit == NULL: dense)
__device__ __forceinline__ float2 pair_flow(const float* __restrict__ flow, const uint8_t* __restrict__ hit, int hit_is,
                                            int hit_tiles_x, int hit_stride, int b, int xx, int yy, int H, int W) {
    if (hit && !tile_covered(hit + (int64_t)b * hit_stride, hit_tiles_x, hit_is, xx, yy)) return make_float2(0.0f, 0.0f);
    return pair_flow(flow, b, xx, yy, H, W);
}

__device__ __forceinline__ DirTaps pair_taps(float2 uv, int xx, int yy, int H, int W) {
    DirTaps d;
    d.uv = uv;
    float ix, iy;
    sample_pos((float)xx, (float)yy, d.uv.x, d.uv.y, W, H, ix, iy);
    d.t = make_taps(ix, iy);
    d.a = pair_addr(d.t, W, H);
    return d;
}

__device__ __forceinline__ void pair_load(const DirTaps& d, const float* __restrict__ src,
                                          const float* __restrict__ tgt, const float* __restrict__ jwarp,
                                          const float* __restrict__ jdirect, int Cj, bool all_jitter_channels,
                                          int b, int64_t pix, int64_t hw, DirRaw2& r) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        r.src[c] = load_quad2(src + ((int64_t)b * 3 + c) * hw, d.a);
        r.tgt[c] = tgt[((int64_t)b * 3 + c) * hw + pix];
    }
    // channels 1, 2 only when the per-channel masks are requested (pair_eval then reads them; otherwise it
    // uses channel 0 three times -- no copies of loaded values here, they would wait for the loads)
    r.jit[0] = load_quad2(jwarp + (int64_t)b * Cj * hw, d.a);
    const F2 z{0.0f, 0.0f};
    r.jit[1].n = z; r.jit[1].s = z; r.jit[2].n = z; r.jit[2].s = z;
    if (all_jitter_channels && Cj == 3) {
        r.jit[1] = load_quad2(jwarp + ((int64_t)b * Cj + 1) * hw, d.a);
        r.jit[2] = load_quad2(jwarp + ((int64_t)b * Cj + 2) * hw, d.a);
    }
    r.jd = jdirect[(int64_t)b * Cj * hw + pix];
}

// ... split for callers that can request the pixel's own values (target, direct jitter) BEFORE the flow is known:
__device__ __forceinline__ void pair_load_own(const float* __restrict__ tgt, const float* __restrict__ jdirect, int Cj, int b,
                                              int64_t pix, int64_t hw, DirRaw2& r) {
#pragma unroll
    for (int c = 0; c < 3; c++) r.tgt[c] = tgt[((int64_t)b * 3 + c) * hw + pix];
    r.jd = jdirect[(int64_t)b * Cj * hw + pix];
}
__device__ __forceinline__ void pair_load_taps(const DirTaps& d, const float* __restrict__ src,
                                               const float* __restrict__ jwarp, int Cj, int b, int64_t hw, DirRaw2& r) {
#pragma unroll
    for (int c = 0; c < 3; c++) r.src[c] = load_quad2(src + ((int64_t)b * 3 + c) * hw, d.a);
    r.jit[0] = load_quad2(jwarp + (int64_t)b * Cj * hw, d.a);
    const F2 z{0.0f, 0.0f};
    r.jit[1].n = z; r.jit[1].s = z; r.jit[2].n = z; r.jit[2].s = z;
}

__device__ __forceinline__ void pin(DirRaw2& r) {
#pragma unroll
    for (int c = 0; c < 3; c++) { pin(r.src[c]); pin(r.jit[c]); pin(r.tgt[c]); }
    pin(r.jd);
}

__device__ __forceinline__ DirRaw unpack(const DirRaw2& r2, const PairAddr& a) {
    DirRaw r;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        r.src[c] = quad_of(r2.src[c], a);
        r.jit[c] = quad_of(r2.jit[c], a);
        r.tgt[c] = r2.tgt[c];
    }
    r.jd = r2.jd;
    return r;
}

__device__ __forceinline__ DirOut pair_eval(const DirTaps& d, const DirRaw& r, int H, int W, float thresh,
                                            bool three_jitter_channels) {
    DirOut o;
    o.m = valid_mask(d.t, W, H, thresh);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        o.s[c] = bilin_q(r.src[c], d.t, d.a) * o.m;
        const float js = bilin_q(three_jitter_channels ? r.jit[c] : r.jit[0], d.t, d.a) * o.m;
        o.wm[c] = o.m * ((js == 1.0f) ? 1.0f : 0.0f);
    }
    o.valid = (o.wm[0] != 0.0f) && (d.uv.x != 0.0f) && (r.jd == 1.0f);
    return o;
}

__device__ __forceinline__ float2 pair_grad(const DirTaps& d, const DirRaw& r, const DirOut& o, int H, int W,
                                             float coef) {
    float gu = 0.0f, gv = 0.0f;
    if (o.valid && coef != 0.0f) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float gix, giy;
            bilin_grad_q(r.src[c], d.t, d.a, gix, giy);
            const float res = o.s[c] - r.tgt[c];
            const float sg = (res > 0.0f) ? 1.0f : ((res < 0.0f) ? -1.0f : 0.0f);
            const float g = sg * coef * o.m;
            gu += g * gix;
            gv += g * giy;
        }
        gu = gu * ((float)W / 2.0f) * (2.0f / (float)max(W - 1, 1));
        gv = gv * ((float)H / 2.0f) * (2.0f / (float)max(H - 1, 1));
    }
    return make_float2(gu, gv);
}

// ---------------------------------------------------------------------------------------
// listed launches: a workgroup's tile of the stacked render's tile list (see warp.hip, "Listed launches")
// ---------------------------------------------------------------------------------------
struct ListArgs {
    const TileList* tlist;
    const uint4* ids;
    unsigned cap;
};

struct TileAt {
    int img, dir, b, tile, x, ry, y;  // image of the stack, direction (0: frame 1's grid, 1: frame 2's), pair, raster tile,
    uint32_t word;                    // pixel column, raster row, image row of this thread; the tile's coverage word
    bool row_covered;                 // this thread's row pair holds a covered pixel (the planes are defined there)
};
__device__ __forceinline__ TileAt tile_at(unsigned gtile, int B, int tiles_x, int T, int is, const uint8_t* __restrict__ hit_lo,
                                          const uint8_t* __restrict__ hit_hi) {
    TileAt t;
    t.img = (int)(gtile / (unsigned)T);
    t.tile = (int)(gtile % (unsigned)T);
    t.dir = t.img >= B ? 1 : 0;
    t.b = t.img - t.dir * B;
    t.word = *reinterpret_cast<const uint32_t*>((t.dir ? hit_hi : hit_lo) + ((int64_t)t.b * T + t.tile) * 4);
    t.x = (t.tile % tiles_x) * 32 + (int)(threadIdx.x & 31u);
    t.ry = (t.tile / tiles_x) * 8 + (int)(threadIdx.x >> 5);
    t.y = is - 1 - t.ry;
    t.row_covered = ((t.word >> (8 * ((t.ry & 7) >> 1))) & 0xffu) != 0u;
    return t;
}

}  // namespace mr
