// mr_common.hpp -- shared device helpers for libmeshraster_hip (gfx950 only).
//
// All parity-critical arithmetic is written one IEEE fp32 operation per C++ operator and
// the library is compiled with -ffp-contract=off, so the results are bit-identical to the
// CPU oracle (oracle/raster_oracle.c) wherever the summation order is the same.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/meshraster_hip.h"

#define MR_WAVE 64

#define MR_CHECK_LAUNCH()                          \
    do {                                           \
        hipError_t e_ = hipGetLastError();         \
        if (e_ != hipSuccess) return (int)e_;      \
    } while (0)

namespace mr {

// Pixel-space bounding box of a face (inclusive); empty when x0 > x1.
struct __attribute__((aligned(8))) FaceBox {
    int16_t x0, x1, y0, y1;
};

// A face as the rasteriser sees it: NDC x,y + metric z per vertex, and the pixel-space
// inverse of [[x0 x1 x2],[y0 y1 y2],[1 1 1]] (upstream kernel forward_face_index_map_1).
struct Face {
    float v[9];
    float inv[9];
};

__device__ __forceinline__ bool backfacing(const float* f) {
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

__device__ __forceinline__ void face_inverse(const float* f, float* inv, int is) {
    const float fis = (float)is;
    float p[3][2];
#pragma unroll
    for (int n = 0; n < 3; n++)
#pragma unroll
        for (int d = 0; d < 2; d++) p[n][d] = 0.5f * (f[3 * n + d] * fis + fis - 1.0f);
    float a[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                  p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                  p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
    const float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) +
                       p[1][0] * (p[2][1] - p[0][1]));
#pragma unroll
    for (int k = 0; k < 9; k++) inv[k] = a[k] / den;
}

__device__ __forceinline__ void load_face(const float* __restrict__ g, Face& f, int is) {
#pragma unroll
    for (int k = 0; k < 9; k++) f.v[k] = g[k];
    face_inverse(f.v, f.inv, is);
}

// Conservative pixel bounding box: every pixel the coverage test can accept is inside.
// Back-facing faces and faces with a NaN coordinate can never win a pixel -> empty.
// Degenerate or infinite faces get the whole screen so that the exact per-pixel test decides, as in
// the brute-force upstream loop: for collinear vertices all three edge functions are the same line
// function, and the pixels lying exactly on that line are accepted along its WHOLE length, also beyond
// the vertices.  "Degenerate" cannot mean a zero pixel-space determinant only: with large coordinates
// that determinant of a mathematically collinear triple rounds to a small non-zero value (found by
// tests/fuzz_parity.py).  The filter uses the two products of the back-face test instead (vertex
// DIFFERENCES first, so a collinear triple gives equal products up to a few ulp): a pixel beyond the
// apex of a sliver passes both long-edge tests only if it is within eps * r of both lines at distance r,
// i.e. only for apex angles of ~1e-7 rad.  Faces whose smallest angle (area / product of the two longest
// edges) is below 16 eps ~ 1e-6 rad are flagged: an order of magnitude of margin, and nothing but
// numerically collinear faces (none of the 28416 front faces of the bench scene).
// The part of face_box that does not depend on the ORDER of the three vertices (a face and its
// fill-back copy -- the same vertices in reverse order -- share it): NaN check, pixel coordinates,
// their finiteness, the clipped bounding box and the product of the two longest edges.  Every value is
// a min / max / |difference| / commutative product of the same operands in both orders, hence
// bit-identical.
struct BoxShared {
    float px[3], py[3];
    float two_longest;
    float tight_need;  // (pa - pb) must exceed this for the tight box to be provably complete (see face_box_orient)
    FaceBox box;       // clipped bbox with a one-pixel margin (empty if off screen)
    FaceBox tight;     // clipped bbox with a 1/16-pixel margin (empty if it holds no pixel centre)
    bool anynan;
    bool nonfinite;    // a pixel coordinate is Inf (or NaN)
};

// Margin of the tight box, in pixels.
#define MR_TIGHT_MARGIN 0.0625f

__device__ __forceinline__ void face_box_shared(const float* f, int is, BoxShared& s) {
    s.anynan = false;
#pragma unroll
    for (int k = 0; k < 9; k++) s.anynan |= (f[k] != f[k]);
    const float fis = (float)is;
#pragma unroll
    for (int n = 0; n < 3; n++) {
        s.px[n] = 0.5f * (f[3 * n + 0] * fis + fis - 1.0f);
        s.py[n] = 0.5f * (f[3 * n + 1] * fis + fis - 1.0f);
    }
    const float e01 = fmaxf(fabsf(f[3] - f[0]), fabsf(f[4] - f[1])), e12 = fmaxf(fabsf(f[6] - f[3]), fabsf(f[7] - f[4])),
                e20 = fmaxf(fabsf(f[0] - f[6]), fabsf(f[1] - f[7]));
    const float emin = fminf(e01, fminf(e12, e20));
    const float emax = fmaxf(e01, fmaxf(e12, e20));
    s.two_longest = emin > 0.0f ? (e01 * e12 * e20) / emin : emax * emax;
    s.nonfinite = false;
#pragma unroll
    for (int n = 0; n < 3; n++) s.nonfinite |= !(fabsf(s.px[n]) <= 3.0e38f) || !(fabsf(s.py[n]) <= 3.0e38f);
    s.box.x0 = 1; s.box.x1 = 0; s.box.y0 = 1; s.box.y1 = 0;
    const float lim = fis - 1.0f;
    const float xlo = floorf(fminf(s.px[0], fminf(s.px[1], s.px[2])));
    const float xhi = ceilf(fmaxf(s.px[0], fmaxf(s.px[1], s.px[2])));
    const float ylo = floorf(fminf(s.py[0], fminf(s.py[1], s.py[2])));
    const float yhi = ceilf(fmaxf(s.py[0], fmaxf(s.py[1], s.py[2])));
    if (!(xhi < 0.0f || yhi < 0.0f || xlo > lim || ylo > lim)) {
        s.box.x0 = (int16_t)fmaxf(xlo, 0.0f);
        s.box.x1 = (int16_t)fminf(xhi, lim);
        s.box.y0 = (int16_t)fmaxf(ylo, 0.0f);
        s.box.y1 = (int16_t)fminf(yhi, lim);
    }
    // The tight box: the pixel centres (integers in this coordinate system) within MR_TIGHT_MARGIN of the vertices'
    // bounding box.  floor(min) .. ceil(max) above keeps everything closer than ONE pixel, i.e. on average two rows and
    // two columns that no pixel of a well-shaped face can be in: for the few-pixel faces of a dense mesh (5.9 x 5.4
    // -> 4.0 x 3.5 pixels on the bench scene) that is more than half of the box, a third of the (face, row) items and
    // a fifth of the (face, tile) records of the tile rasteriser.  Which faces may use it: face_box_orient.
    s.tight.x0 = 1; s.tight.x1 = 0; s.tight.y0 = 1; s.tight.y1 = 0;
    {
        const float m = MR_TIGHT_MARGIN;
        const float txlo = fmaxf(ceilf(fminf(s.px[0], fminf(s.px[1], s.px[2])) - m), 0.0f);
        const float txhi = fminf(floorf(fmaxf(s.px[0], fmaxf(s.px[1], s.px[2])) + m), lim);
        const float tylo = fmaxf(ceilf(fminf(s.py[0], fminf(s.py[1], s.py[2])) - m), 0.0f);
        const float tyhi = fminf(floorf(fmaxf(s.py[0], fmaxf(s.py[1], s.py[2])) + m), lim);
        if (txlo <= txhi && tylo <= tyhi) {
            s.tight.x0 = (int16_t)txlo; s.tight.x1 = (int16_t)txhi;
            s.tight.y0 = (int16_t)tylo; s.tight.y1 = (int16_t)tyhi;
        }
    }
    // 64 u (emax is / margin) x (product of the two longest edges), u = 2^-24; +Inf when a vertex lies more than four
    // screens away (the bound below assumes |pixel - vertex| <= ~8.5 is)
    float far_out = 0.0f;
#pragma unroll
    for (int n = 0; n < 3; n++) far_out = fmaxf(far_out, fmaxf(fabsf(s.px[n]), fabsf(s.py[n])));
    s.tight_need = (far_out <= 4.0f * fis) ? (64.0f * 5.9604645e-8f / MR_TIGHT_MARGIN) * (emax * fis) * s.two_longest
                                            : __builtin_inff();
}

// The order-dependent part, for the vertex order (a, b, c) = (0, 1, 2) or, REV, (2, 1, 0).
template <bool REV>
__device__ __forceinline__ FaceBox face_box_orient(const float* f, int is, const BoxShared& s) {
    constexpr int A = REV ? 2 : 0, B = 1, C = REV ? 0 : 2;
    FaceBox b;
    b.x0 = 1; b.x1 = 0; b.y0 = 1; b.y1 = 0;
    const float xa = f[3 * A], ya = f[3 * A + 1], xb = f[3 * B], yb = f[3 * B + 1], xc = f[3 * C], yc = f[3 * C + 1];
    // backfacing(): (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])
    const float pa = (yc - ya) * (xb - xa), pb = (yb - ya) * (xc - xa);
    if (s.anynan || pa < pb) return b;
    const float den = (s.px[C] * (s.py[A] - s.py[B]) + s.px[A] * (s.py[B] - s.py[C]) + s.px[B] * (s.py[C] - s.py[A]));
    const bool full = !(den != 0.0f) || !(fabsf(den) <= 3.0e38f) || !(pa - pb > 16.0f * 5.9604645e-8f * s.two_longest) ||
                      s.nonfinite;
    if (full) {
        b.x0 = 0; b.x1 = (int16_t)(is - 1); b.y0 = 0; b.y1 = (int16_t)(is - 1);
        return b;
    }
    // Tight box.  A pixel p the three fp32 edge tests accept lies, for every edge (S -> Q), on the inner side of the
    // edge's line or within an ANGLE of 6 u of it seen from S (each side of `(yp - Sy)(Qx - Sx) < (xp - Sx)(Qy - Sy)`
    // is two roundings of exact differences away from a product of magnitude <= |p - S| |Q - S|).  Take the vertex M of
    // largest x: the face's wedge at M lies in x <= Mx, so a pixel with x > Mx is outside one of M's two edges, hence
    // within 6 u |p - S| of that edge's line; either it sits next to the edge itself -- then x - Mx <= 6 u |p - S|
    // -- or on the line's extension beyond M, at distance r from M, where it is r sin(angle at M) outside M's OTHER
    // edge, which the second test forgives only if r sin(angle) <= 6 u (r + L): r <= 6 u L / (sin(angle) - 6 u).  With
    // sin(smallest angle) >= (pa - pb) / (2 x product of the two longest edges) (max-norm lengths) and L <= sqrt(2)
    // emax is / 2 pixels, `pa - pb > tight_need` gives r < margin / 7, and |p - S| <= 8.5 is bounds the first case
    // by 3e-6 is pixels: every accepted pixel is within the margin of the bounding box for rasters up to 16384.
    // The same holds for the other three sides.  Thin faces (smallest angle below ~1e-4 rad at 256 pixels) keep the
    // one-pixel box.
    if (pa - pb > s.tight_need) return s.tight;
    return s.box;
}

__device__ __forceinline__ FaceBox face_box(const float* f, int is) {
    BoxShared s;
    face_box_shared(f, is, s);
    return face_box_orient<false>(f, is, s);
}

// Barycentric weights (clamped, normalised) and perspective-correct depth of pixel (xi, yi)
// on face f -- the arithmetic of the upstream face loop after the inside test.
__device__ __forceinline__ void bary(const Face& f, int xi, int yi, float& zp, float* w) {
    const float fx = (float)xi, fy = (float)yi;
    w[0] = f.inv[0] * fx + f.inv[1] * fy + f.inv[2];
    w[1] = f.inv[3] * fx + f.inv[4] * fy + f.inv[5];
    w[2] = f.inv[6] * fx + f.inv[7] * fy + f.inv[8];
    float ws = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        w[k] = fminf(fmaxf(w[k], 0.0f), 1.0f);
        ws += w[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] /= ws;
    zp = 1.0f / (w[0] / f.v[2] + w[1] / f.v[5] + w[2] / f.v[8]);
}

// ---------------------------------------------------------------------------------------------------------------
// IEEE divisions that share work.  Two of every five vector instructions of the forward tile kernel used to be the
// twelve-instruction sequence the compiler emits for a correctly rounded fp32 `/` (v_div_scale x 2, v_rcp_f32, six
// fma / mul, v_div_fmas, v_div_fixup): nine divisions by ONE determinant per face set-up, three by one weight sum and
// three by the face's three vertex depths per fragment, the same again per resolved pixel.  Without the scaling
// steps (which only exist to keep intermediates of extreme operands out of the denormal / overflow range) that
// sequence is: y = refined reciprocal of b (rcp + one Newton step), then q0 = a y, two residual corrections q <- q +
// (a - b q) y.  The residual a - b q is exact in fp32 (one fma) as long as nothing underflows, and the corrected
// quotient is the correctly rounded a / b -- the SAME bits as `/` and as the CPU oracle's division -- for every
// operand pair whose magnitudes keep the intermediates normal.  y depends on b alone, so divisions by a common b share
// it: 3 + 5 n instead of 12 n instructions.  (CPU emulation with the reciprocal perturbed by +-1 ulp: 12 M random and
// adversarial pairs, 0 mismatches; on the GPU tests/test_gpu_raster.py compares the tile kernel with these paths on
// and off bit for bit, and mr_selftest_division the primitives themselves against `/`.)
// Range discipline: `division_safe_face` (evaluated once per face by the per-face pass) admits a face to the fast
// paths only if its pixel coordinates are 0 or within [2^-24, 2^24], its determinant and its three depths within
// [2^-30, 2^30]: then every numerator of the set-up is 0 or within [2^-71, 2^49] and every inverse entry 0 or above
// 2^-101, with residuals that are multiples of 2^-118 or more.  Per fragment, the weight sum must be at least 2^-30 and
// each clamped weight 0 or at least 2^-60 (the quotients are then 0 or at least 2^-62, their residuals multiples of
// 2^-109), per resolved pixel the depth within [2^-30, 2^30].  Everything else takes the plain `/` path.
__device__ __forceinline__ float rcp_refined(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
// a / b correctly rounded, y = rcp_refined(b); operands within the ranges stated above
__device__ __forceinline__ float div_refined(float a, float b, float y) {
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    // (a zero numerator: the corrections add a +0 residual to q0 = -0 and lose the sign IEEE gives the quotient
    // (-0 / b = -0 for b > 0); q0 = a y always carries the right one)
    return __builtin_copysignf(__builtin_fmaf(r1, y, q1), q0);
}
// |x| within [2^lo, 2^hi] (false for NaN / Inf / zero / denormals)
__device__ __forceinline__ bool mag_within(float x, int lo, int hi) {
    const uint32_t u = __float_as_uint(x) & 0x7fffffffu;
    return u - ((uint32_t)(lo + 127) << 23) <= (((uint32_t)(hi + 127) << 23) - ((uint32_t)(lo + 127) << 23));
}
// x == 0 or |x| within [2^lo, 2^hi]
__device__ __forceinline__ bool zero_or_mag_within(float x, int lo, int hi) {
    return (__float_as_uint(x) & 0x7fffffffu) == 0u || mag_within(x, lo, hi);
}

__device__ __forceinline__ bool division_safe_face(const float* f, int is) {
    const float fis = (float)is;
    float p[3][2];
    bool ok = true;
#pragma unroll
    for (int n = 0; n < 3; n++) {
#pragma unroll
        for (int d = 0; d < 2; d++) {
            p[n][d] = 0.5f * (f[3 * n + d] * fis + fis - 1.0f);
            ok = ok && zero_or_mag_within(p[n][d], -24, 24);
        }
        ok = ok && mag_within(f[3 * n + 2], -30, 30);
    }
    const float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]));
    return ok && mag_within(den, -30, 30);
}

// face_inverse for division-safe faces: the nine divisions by the determinant share one reciprocal (bit-identical)
__device__ __forceinline__ void face_inverse_shared(const float* f, float* inv, int is) {
    const float fis = (float)is;
    float p[3][2];
#pragma unroll
    for (int n = 0; n < 3; n++)
#pragma unroll
        for (int d = 0; d < 2; d++) p[n][d] = 0.5f * (f[3 * n + d] * fis + fis - 1.0f);
    float a[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                  p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                  p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
    const float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) +
                       p[1][0] * (p[2][1] - p[0][1]));
    const float y = rcp_refined(den);
#pragma unroll
    for (int k = 0; k < 9; k++) inv[k] = div_refined(a[k], den, y);
}

// bary() for fragments of division-safe faces; yz[k] = rcp_refined(f.v[3 k + 2]).  Returns false -- with zp and w
// untouched -- when this fragment's weights are outside the fast range: the caller then uses bary().
__device__ __forceinline__ bool bary_shared(const Face& f, const float* yz, int xi, int yi, float& zp, float* w) {
    const float fx = (float)xi, fy = (float)yi;
    float t[3];
    t[0] = f.inv[0] * fx + f.inv[1] * fy + f.inv[2];
    t[1] = f.inv[3] * fx + f.inv[4] * fy + f.inv[5];
    t[2] = f.inv[6] * fx + f.inv[7] * fy + f.inv[8];
    float ws = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        t[k] = fminf(fmaxf(t[k], 0.0f), 1.0f);
        ws += t[k];
    }
    // each weight 0 or >= 2^-60 (they are non-negative: u - 1 wraps for +0), the sum >= 2^-30
    const uint32_t lo = min(min(__float_as_uint(t[0]) - 1u, __float_as_uint(t[1]) - 1u), __float_as_uint(t[2]) - 1u);
    if (!(lo >= ((uint32_t)(127 - 60) << 23) - 1u && ws >= 9.313225746154785e-10f)) return false;
    const float y = rcp_refined(ws);
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = div_refined(t[k], ws, y);
    zp = 1.0f / (div_refined(w[0], f.v[2], yz[0]) + div_refined(w[1], f.v[5], yz[1]) + div_refined(w[2], f.v[8], yz[2]));
    return true;
}

// Coverage + barycentric weights + perspective-correct depth of one pixel against one
// front-facing face (upstream kernel forward_face_index_map_2, body of the face loop).
__device__ __forceinline__ bool cover(const Face& f, int xi, int yi, int is, float near_, float far_,
                                      float& zp, float* w) {
    const float fis = (float)is;
    const float yp = (float)(2 * yi + 1 - is) / fis;
    const float xp = (float)(2 * xi + 1 - is) / fis;
    const float* v = f.v;
    if (((yp - v[1]) * (v[3] - v[0]) < (xp - v[0]) * (v[4] - v[1])) ||
        ((yp - v[4]) * (v[6] - v[3]) < (xp - v[3]) * (v[7] - v[4])) ||
        ((yp - v[7]) * (v[0] - v[6]) < (xp - v[6]) * (v[1] - v[7])))
        return false;
    bary(f, xi, yi, zp, w);
    // (a NaN depth passes upstream's near / far test and then loses `zp < depth`: not covered)
    if (!(zp > near_ && zp < far_)) return false;
    return true;
}

// Texel layout of the 2x2x2 vertex-colour texture (libyana's batch_vertex_textures -- source absent): the
// texel with a 1 on axis a -- (1,0,0), (0,1,0), (0,0,1) -- holds the colour of vertex sigma(a) of the face.  `code`
// packs sigma two bits per axis (MR_TEXEL_LAYOUT_DEFAULT = identity: what utils/textutils.py assumes); 0 = default.
// For the reversed copy of a face (fill-back: vertices in reverse order, texture axes mirrored, renderer.py:250-252) the
// same texel holds the colour of the copy's vertex 2 - sigma(2 - a).  Sampling coordinate a is the barycentric weight of
// geometry vertex a, so this table is all that ties geometry to colours in the fused vertex-colour kernels.
__device__ __forceinline__ int texel_vertex(int code, int axis, bool reversed) {
    const int a = reversed ? 2 - axis : axis;
    const int s = ((code ? code : MR_TEXEL_LAYOUT_DEFAULT) >> (2 * a)) & 3;
    return reversed ? 2 - s : s;
}
static inline bool texel_layout_ok(int code) {  // 0 (default) or a permutation of {0, 1, 2}, two bits per axis
    if (code == 0) return true;
    const int a = code & 3, b = (code >> 2) & 3, c = (code >> 4) & 3;
    return (code >> 6) == 0 && a < 3 && b < 3 && c < 3 && a != b && a != c && b != c;
}
__device__ __forceinline__ int sel3(int v0, int v1, int v2, int s) { return s == 0 ? v0 : (s == 1 ? v1 : v2); }

// Order-preserving map float -> uint32 (so that an unsigned min is a float min).
__device__ __forceinline__ uint32_t f2ord(float x) {
    uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Texture-space coordinates of a hit pixel (upstream forward_texture_sampling).
__device__ __forceinline__ void tex_coords(const float* w, float depth, const float* v, int ts,
                                           float eps, float* tif) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t = w[k] * (float)(ts - 1) * (depth / v[3 * k + 2]);
        t = fmaxf(t, 0.0f);
        t = fminf(t, (float)(ts - 1) - eps);
        tif[k] = t;
    }
}

__device__ __forceinline__ void tex_tap(const float* tif, int pn, int ts, float& wgt, int& isc) {
    wgt = 1.0f;
    int ti[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int fl = (int)tif[k];
        if (((pn >> k) & 1) == 0) {
            wgt *= 1.0f - (tif[k] - (float)fl);
            ti[k] = fl;
        } else {
            wgt *= tif[k] - (float)fl;
            ti[k] = fl + 1;
        }
    }
    isc = ti[0] * ts * ts + ti[1] * ts + ti[2];
}

// axis-angle -> rotation through the normalised quaternion, the operation order of batch_rodrigues
__device__ __forceinline__ void rodrigues(const float* r, float* R, float* q_out) {
    const float a0 = r[0] + 1e-8f, a1 = r[1] + 1e-8f, a2 = r[2] + 1e-8f;
    const float angle = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    const float half = angle * 0.5f;
    const float c = cosf(half), s = sinf(half);
    float q[4] = {c, s * (r[0] / angle), s * (r[1] / angle), s * (r[2] / angle)};
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = q[k] / n;
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;     R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;     R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
    if (q_out) {
#pragma unroll
        for (int k = 0; k < 4; k++) q_out[k] = q[k];
    }
}

// adjoint of rodrigues: gR[9] -> gr[3]
__device__ __forceinline__ void rodrigues_bwd(const float* r, const float* gR, float* gr) {
    // recompute the forward intermediates
    const float a0 = r[0] + 1e-8f, a1 = r[1] + 1e-8f, a2 = r[2] + 1e-8f;
    const float angle = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    const float half = angle * 0.5f;
    const float c = cosf(half), s = sinf(half);
    const float ax[3] = {r[0] / angle, r[1] / angle, r[2] / angle};
    const float p[4] = {c, s * ax[0], s * ax[1], s * ax[2]};  // un-normalised quaternion
    const float n = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
    const float q[4] = {p[0] / n, p[1] / n, p[2] / n, p[3] / n};
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    // R(q) -> gq
    float gq[4];
    gq[0] = 2 * (w * (gR[0] + gR[4] + gR[8]) + x * (gR[7] - gR[5]) + y * (gR[2] - gR[6]) + z * (gR[3] - gR[1]));
    gq[1] = 2 * (x * (gR[0] - gR[4] - gR[8]) + y * (gR[1] + gR[3]) + z * (gR[2] + gR[6]) + w * (gR[7] - gR[5]));
    gq[2] = 2 * (y * (gR[4] - gR[0] - gR[8]) + x * (gR[1] + gR[3]) + z * (gR[5] + gR[7]) + w * (gR[2] - gR[6]));
    gq[3] = 2 * (z * (gR[8] - gR[0] - gR[4]) + x * (gR[2] + gR[6]) + y * (gR[5] + gR[7]) + w * (gR[3] - gR[1]));
    // q = p / |p|
    const float dot = gq[0] * q[0] + gq[1] * q[1] + gq[2] * q[2] + gq[3] * q[3];
    float gp[4];
#pragma unroll
    for (int k = 0; k < 4; k++) gp[k] = (gq[k] - q[k] * dot) / n;
    // p = (cos h, sin h * axis), h = angle / 2, axis = r / angle, angle = |r + 1e-8|
    const float g_half = -s * gp[0] + c * (gp[1] * ax[0] + gp[2] * ax[1] + gp[3] * ax[2]);
    const float g_ax[3] = {s * gp[1], s * gp[2], s * gp[3]};
    float g_angle = 0.5f * g_half;
#pragma unroll
    for (int k = 0; k < 3; k++) g_angle -= g_ax[k] * r[k] / (angle * angle);
    const float a[3] = {a0, a1, a2};
#pragma unroll
    for (int k = 0; k < 3; k++) gr[k] = g_ax[k] / angle + g_angle * a[k] / angle;
}

// Covered-tile lists of the 2B stack images of a frame pair (round 5, ABI 7): mr_flow_pair_forward_grad_tiles' finalize
// launch compacts the coverage words it reads anyway, mr_flow_pair_backward_unit_tiles hands out its workgroups over them in
// proportion to the images' covered tiles (raster_bwd.hip, scatter_tiles_body WORK).  Layout of the buffer:
//   int n_cov[2B] (padded to 256 bytes) | uint16 cov[2B][tiles per image]  (ids of the image's covered tiles, ascending)
struct ScatterWork {
    int* n_cov;
    unsigned short* cov;
};
__host__ __device__ inline int64_t scatter_work_cov_offset(int images) { return ((int64_t)images * 4 + 255) & ~(int64_t)255; }
__host__ __device__ inline int64_t scatter_work_bytes(int images, int tiles) {
    return scatter_work_cov_offset(images) + (((int64_t)images * tiles * 2 + 255) & ~(int64_t)255);
}
__host__ __device__ inline ScatterWork scatter_work_at(void* buf, int images) {
    ScatterWork w;
    w.n_cov = reinterpret_cast<int*>(buf);
    w.cov = buf ? reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(buf) + scatter_work_cov_offset(images)) : nullptr;
    return w;
}

// inclusive sum over the lanes of a wave (int): row shifts and the two row broadcasts of the gfx9 DPP unit -- six vector
// instructions, no LDS crossbar (a lane without a source adds 0).  Call it with ALL lanes of the wave active.
__device__ __forceinline__ int wave_incl_sum(int v, int lane = 0) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
    return v;
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8; give every XCD a
// contiguous range of logical ids so the tiles / faces of one image share one L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    if ((nblocks & 7u) != 0u) return bid;
    return (bid & 7u) * (nblocks >> 3) + (bid >> 3);
}

// Header of the tile list of a render launch (one per call, in the render's workspace; raster_fwd.hip builds it, the
// listed warp kernels of warp.hip walk it too).  Entries are uint4 {global tile id = image * tiles per image + tile (raster
// orientation, row-major), offset and count of the bin's records, length of the image's large list}.
struct TileList {
    unsigned n_heavy;  // tiles with at least HEAVY_RECS candidate records, all images: dispatched FIRST; entries [0, n_heavy)
    unsigned n_light;  // the other tiles with at least one candidate record; entries [cap, cap + n_light)
    unsigned n_bg;     // dense launches: the tiles WITHOUT candidates (pure background), listed in a separate id array
    unsigned pad[61];
};

// A workgroup's place in a listed launch.  The dispatcher puts workgroup i on XCD i % 8, and every XCD has its own L2:
// each XCD gets a contiguous eighth of the list's heavy part followed by the same eighth of its light part (the entries
// of an image are contiguous in both parts, so an image's records stay in one L2, every XCD gets the same share of the
// heavy tiles, and within an XCD the heavy ones are dispatched first).  Local entry j of XCD x is list entry slot(j).
struct ListSlice {
    unsigned first_heavy, n_heavy, first_light, n_local, stride;
    __device__ __forceinline__ unsigned slot(unsigned j, unsigned cap) const {
        return j < n_heavy ? first_heavy + j : cap + first_light + (j - n_heavy);
    }
};
__device__ __forceinline__ ListSlice list_slice(unsigned n_heavy, unsigned n_light, unsigned& j) {
    const unsigned nx = (gridDim.x & 7u) ? 1u : 8u;  // (grids that are no multiple of 8: one slice)
    const unsigned x = blockIdx.x % nx;
    j = blockIdx.x / nx;
    ListSlice s;
    s.first_heavy = (unsigned)((unsigned long long)n_heavy * x / nx);
    s.n_heavy = (unsigned)((unsigned long long)n_heavy * (x + 1) / nx) - s.first_heavy;
    s.first_light = (unsigned)((unsigned long long)n_light * x / nx);
    s.n_local = s.n_heavy + (unsigned)((unsigned long long)n_light * (x + 1) / nx) - s.first_light;
    s.stride = gridDim.x / nx;
    return s;
}

}  // namespace mr
