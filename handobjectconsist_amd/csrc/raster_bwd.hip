// raster_bwd.hip -- backward of the rasteriser for gfx950 (MI355X).
//
// Replaces upstream kernels backward_pixel_map / backward_textures / backward_depth_map
// (/root/reference/meshreg/neurender/rasterize.py:269-315).
//
// Kernels, in file order:
//   * gather_kernel<IMG,TEX,DEPTH> -- E + F for generic 2x2x2 textures: FACE-PARALLEL GATHER instead of
//     per-pixel global atomics.  Four lanes share a face (one bbox row each), walk its pixel bounding
//     box in face_index_map and, for the pixels it won, recompute barycentrics / sampling weights from
//     the vertices (nothing but face_index_map is read back from the forward) into register
//     accumulators; faces with a very large bbox are walked by the whole wave.  Every output element
//     is written exactly once: no memset, no atomics, deterministic.
//   * gather_vc_kernel -- the same idea for the vertex-colour mode with an LDS fragment ring (probe,
//     compact the won pixels, shade with full lanes); kept as the fallback for meshes whose colour
//     table does not fit LDS.
//   * scatter_vc_kernel<STORED> -- E for the vertex-colour mode, the training path: PIXEL-parallel,
//     block-private [V,3] table in LDS in 64-bit fixed point, optionally on the forward's stored
//     weight / depth maps (see the comment at the kernel).
//   * textures_atomic_*, depth_atomic_stored_kernel -- generic texture size / upstream-compatible
//     variants (per-pixel atomics on stored or recomputed sampling weights) behind the five-entry-point API.
//   * pixel_map_kernel<IMG> -- kernel D as upstream walks it, one wave per face, reading the eight
//     planes in place (reference algorithm; no workspace needed).
//   * pixel_map_strip_kernel<IMG, L> -- kernel D by strips of image lines staged in LDS, fed by per-image owner
//     records from compact_owners_kernel (the default when a workspace is given; see the comment there).
#include "mr_common.hpp"
#include "warp_device.hpp"
#include <algorithm>
#include <type_traits>

namespace mr {


// ---------------------------------------------------------------------------------------
// map accessors: IMG = image orientation (vertically flipped, NCHW rgb), else raster NHWC
// ---------------------------------------------------------------------------------------
template <bool IMG>
__device__ __forceinline__ int64_t idx1(int b, int yi, int xi, int is) {
    return IMG ? ((int64_t)b * is + (is - 1 - yi)) * is + xi : ((int64_t)b * is + yi) * is + xi;
}
template <bool IMG>
__device__ __forceinline__ int64_t idx3(int b, int yi, int xi, int c, int is) {
    return IMG ? (((int64_t)b * 3 + c) * is + (is - 1 - yi)) * is + xi
               : (((int64_t)b * is + yi) * is + xi) * 3 + c;
}

// ---------------------------------------------------------------------------------------
// E + F: face-parallel gather (texture size 2)
// ---------------------------------------------------------------------------------------
struct GatherParams {
    const float* faces;
    const int32_t* fim;       // raster orientation
    const float* grad_rgb;    // nullable
    const float* grad_depth;  // nullable
    float* grad_faces;        // nullable
    float* grad_textures;     // nullable (ts == 2 only)
    int B, F, is;
    float eps;
    int accumulate_faces;     // 1: grad_faces already holds the pixel-map term
    // optional list of the faces that own a pixel (compact_owners_kernel): only those are walked, the rows of the
    // others were zeroed when the list was built
    const uint32_t* owners;
    const unsigned* n_owners;
    int dbg_rows;             // profiling: 1 = the row-walking form of the small-face path
};

template <bool IMG, bool TEX, bool DEPTH>
__device__ __forceinline__ void gather_pixel(const GatherParams& p, const Face& f, int b, int xi, int yi,
                                             float* gt, float* gf) {
    float w[3], zp;
    bary(f, xi, yi, zp, w);
    const int is = p.is;
    if (TEX) {
        float tif[3];
        tex_coords(w, zp, f.v, 2, p.eps, tif);
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; c++) g[c] = p.grad_rgb[idx3<IMG>(b, yi, xi, c, is)];
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            // ts == 2 and 0 <= tif < 1: floor is 0, the tap index is the bit pattern of pn
            float wg = 1.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) wg *= ((pn >> k) & 1) ? (tif[k] - 0.0f) : (1.0f - (tif[k] - 0.0f));
            const int isc = ((pn & 1) << 2) | (((pn >> 1) & 1) << 1) | ((pn >> 2) & 1);
#pragma unroll
            for (int c = 0; c < 3; c++) gt[isc * 3 + c] += wg * g[c];
        }
    }
    if (DEPTH) {
        const float gd = p.grad_depth[idx1<IMG>(b, yi, xi, is)];
        const float d2 = zp * zp;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float zk = f.v[3 * k + 2];
            gf[3 * k + 2] += gd * w[k] * d2 / (zk * zk);
        }
        float tmp[2] = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int l = 0; l < 3; l++) tmp[k] += -f.inv[3 * l + k] / f.v[3 * l + 2];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int l = 0; l < 2; l++) gf[3 * k + l] += -gd * tmp[l] * w[k] * d2 * (float)is / 2.0f;
    }
}

// GGL (generic gather) / GLPF (vertex-colour gather) lanes per face: lane `sub` walks bbox rows sub, sub + lanes, ...;
// the partial sums are combined with shuffles inside the lane group (fixed order: deterministic).  Faces whose bbox
// exceeds GATHER_BIG pixels are walked by the whole wave instead.
constexpr int GLPF = 4;
#ifndef MR_GGL
#define MR_GGL 8
#endif
constexpr int GGL = MR_GGL;   // lanes per face of the generic gather (E: 74 -> 64 us, E + F: 105 -> 93 us against 4; 16: 70 / 101)
constexpr int GATHER_BIG = 128;          // vertex-colour gather: bbox area above which the whole wave probes
// (256 until round 5: the 114 faces of the bench scene with boxes of 256-500 pixels then took the whole-wave path ONE BEHIND THE
// OTHER inside the wave that holds them -- consecutive wrist faces -- while as ordinary lane groups they advance side by side: E
// alone 93.7 -> 90.4 us at 256 x 256, 263 -> 230 at 640 x 640; 8192: no further gain at 256, slower at 640)
#ifndef MR_GATHER_BIG
#define MR_GATHER_BIG 1024
#endif
constexpr int GATHER_BIG_GENERIC = MR_GATHER_BIG;  // generic gather (direct accumulation on the wave-cooperative path)

template <bool TEX, bool DEPTH>
__device__ __forceinline__ void gather_store(const GatherParams& p, int64_t i, const float* gt, const float* gf) {
    if (TEX) {
        float4* o = reinterpret_cast<float4*>(p.grad_textures + i * 24);
#pragma unroll
        for (int k = 0; k < 6; k++) o[k] = make_float4(gt[4 * k], gt[4 * k + 1], gt[4 * k + 2], gt[4 * k + 3]);
    }
    if (p.grad_faces) {
        if (p.accumulate_faces) {
            // (kernel D's sums are in the row already: ADD to them with fire-and-forget atomics -- a load + add + store would
            // put one more dependent round trip at the end of every wave; one writer per element here, the order of the two
            // kernels' contributions is the stream's)
            if (DEPTH)
#pragma unroll
                for (int k = 0; k < 9; k++)
                    if (gf[k] != 0.0f) unsafeAtomicAdd(&p.grad_faces[i * 9 + k], gf[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 9; k++) p.grad_faces[i * 9 + k] = DEPTH ? gf[k] : 0.0f;
        }
    }
}

#ifdef MR_GATHER_WPE
#define MR_GATHER_ATTR __attribute__((amdgpu_waves_per_eu(MR_GATHER_WPE, MR_GATHER_WPE)))
#else
#define MR_GATHER_ATTR
#endif
// (`vbid` of `vgrid`: the workgroup's place in the gather's own grid -- blockIdx.x of gridDim.x when it is launched alone,
// a virtual index when its workgroups are interleaved with kernel D's strips in one launch, strip_gather_kernel)
template <bool IMG, bool TEX, bool DEPTH>
__device__ __forceinline__ void gather_body(const GatherParams& p, const unsigned vbid, const unsigned vgrid) {
    constexpr int NT = TEX ? 24 : 1, NF = DEPTH ? 9 : 1;
    // (owner list: the entries in use are the first ones -- plain block order, so that they spread over the XCDs)
    const int64_t gid = (int64_t)(p.owners ? vbid : xcd_remap(vbid, vgrid)) * blockDim.x + threadIdx.x;
    const int64_t slot = gid / GGL;
    const int sub = (int)(gid % GGL);
    const int lane = threadIdx.x & 63;
    // (the list entry is requested TOGETHER with the list's length, not behind the test on it: the list has room for every
    // face, the entry is inside the allocation whatever the length turns out to be -- one dependent round trip less)
    uint32_t own = 0u;
    if (p.owners) own = p.owners[slot];
    const int64_t total = p.owners ? (int64_t)*p.n_owners : (int64_t)p.B * p.F;
    asm volatile("" : "+v"(own));  // (keeps the load in front of the branch)
    if ((gid - threadIdx.x) / GGL >= total) return;  // block-uniform: nothing left in the list
    const bool valid = slot < total;
    const int64_t i = p.owners ? (valid ? (int64_t)own : 0) : slot;
    const int b = valid ? (int)(i / p.F) : 0;
    const int fn = valid ? (int)(i % p.F) : 0;
    const int is = p.is;

    FaceBox bx;
    bx.x0 = 1; bx.x1 = 0; bx.y0 = 1; bx.y1 = 0;
    if (valid) {
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = p.faces[i * 9 + k];
        bx = face_box(v, is);
    }
    const bool nonempty = valid && bx.x0 <= bx.x1;
    const int bw = bx.x1 - bx.x0 + 1, bh = bx.y1 - bx.y0 + 1;
    const bool big = nonempty && bw * bh > GATHER_BIG_GENERIC;

    float gt[NT], gf[NF];

    // very large faces first: the whole wave walks the bbox, lane per pixel, butterfly-reduce,
    // the owner lane stores the result straight away
    unsigned long long m_big = (p.dbg_rows & 2) ? 0ull : __ballot(big && sub == 0);  // (profiling: 2 = big faces skipped)
    while (m_big) {
        const int src = __ffsll((long long)m_big) - 1;
        m_big &= m_big - 1;
        const int x0 = __shfl((int)bx.x0, src), y0 = __shfl((int)bx.y0, src);
        const int w_ = __shfl(bw, src), n = w_ * __shfl(bh, src);
        const int bb = __shfl(b, src), ff = __shfl(fn, src);
        Face fb;
        load_face(p.faces + ((int64_t)bb * p.F + ff) * 9, fb, is);
        const int32_t* fim_b = p.fim + (int64_t)bb * is * is;
#pragma unroll
        for (int k = 0; k < NT; k++) gt[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < NF; k++) gf[k] = 0.0f;
        // lane per column (chunks of 64 columns), rows in groups of 8 with the 8 index-map loads
        // issued back to back (a degenerate face carries a full-screen bbox: 1024 probes per lane)
        (void)n;
        const int h_ = __shfl(bh, src);
        for (int cx = lane; cx < w_; cx += MR_WAVE) {
            const int xi = x0 + cx;
            for (int r = 0; r < h_; r += 8) {
                int hit[8];
#pragma unroll
                for (int u = 0; u < 8; u++) hit[u] = fim_b[min(y0 + r + u, y0 + h_ - 1) * is + xi];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (r + u < h_ && hit[u] == ff) gather_pixel<IMG, TEX, DEPTH>(p, fb, bb, xi, y0 + r + u, gt, gf);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            if (TEX)
#pragma unroll
                for (int k = 0; k < NT; k++) gt[k] += __shfl_xor(gt[k], off);
            if (DEPTH)
#pragma unroll
                for (int k = 0; k < NF; k++) gf[k] += __shfl_xor(gf[k], off);
        }
        if (lane == src) gather_store<TEX, DEPTH>(p, (int64_t)bb * p.F + ff, gt, gf);
    }

#pragma unroll
    for (int k = 0; k < NT; k++) gt[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < NF; k++) gf[k] = 0.0f;
    if (nonempty && !big) {
        Face f;
        load_face(p.faces + i * 9, f, is);
        const int32_t* fim_b = p.fim + (int64_t)b * is * is;
        if (p.dbg_rows) {  // (profiling: the round-4 form -- every lane evaluates the pixels of its own rows)
            for (int yi = bx.y0 + sub; yi <= bx.y1; yi += GGL)
                for (int xi = bx.x0; xi <= bx.x1; xi += 4) {
                    int hit[4];  // four probes in flight
#pragma unroll
                    for (int u = 0; u < 4; u++) hit[u] = fim_b[yi * is + min(xi + u, (int)bx.x1)];
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (xi + u <= bx.x1 && hit[u] == fn) gather_pixel<IMG, TEX, DEPTH>(p, f, b, xi + u, yi, gt, gf);
                }
        } else {
            // Round 5.  The face's box goes by chunks of GGL rows x 4 columns: lane `sub` probes row `sub` of the chunk (four
            // probes in flight), the group ORs its lanes' results into one 32-bit mask, and the pixels the face WON are then
            // dealt out over the group's lanes by rank -- the few-pixel faces of a dense mesh win 4.6 of the 14 pixels of
            // their box, so one trip through the ~250 instructions of a pixel serves a whole face at more than half of the
            // lanes, where "every lane walks its own row" ran four trips (one per column) at a sixth of them.
            for (int yc = bx.y0; yc <= bx.y1; yc += GGL)
                for (int xc = bx.x0; xc <= bx.x1; xc += 4) {
                    const int yi = yc + sub;
                    const int yl = min(yi, (int)bx.y1);
                    int hit[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) hit[u] = fim_b[yl * is + min(xc + u, (int)bx.x1)];
                    unsigned gm = 0u;
#pragma unroll
                    for (int u = 0; u < 4; u++) gm |= (yi <= bx.y1 && xc + u <= bx.x1 && hit[u] == fn) ? (1u << (sub * 4 + u)) : 0u;
#pragma unroll
                    for (int off = 1; off < GGL; off <<= 1) gm |= (unsigned)__shfl_xor((int)gm, off);
                    const int won = __popc(gm);
                    for (int r = sub; r < won; r += GGL) {
                        unsigned m = gm;
                        for (int k = 0; k < r; k++) m &= m - 1u;  // drop the r lowest set bits
                        const int pos = __ffs((int)m) - 1;
                        gather_pixel<IMG, TEX, DEPTH>(p, f, b, xc + (pos & 3), yc + (pos >> 2), gt, gf);
                    }
                }
        }
    }
    // quad reduction: afterwards every lane of the quad holds the face's sums
#pragma unroll
    for (int off = 1; off < GGL; off <<= 1) {
        if (TEX)
#pragma unroll
            for (int k = 0; k < NT; k++) gt[k] += __shfl_xor(gt[k], off);
        if (DEPTH)
#pragma unroll
            for (int k = 0; k < NF; k++) gf[k] += __shfl_xor(gf[k], off);
    }
    if (valid && sub == 0 && !big) gather_store<TEX, DEPTH>(p, i, gt, gf);
}
template <bool IMG, bool TEX, bool DEPTH>
__global__ void __launch_bounds__(256) MR_GATHER_ATTR gather_kernel(GatherParams p) {
    gather_body<IMG, TEX, DEPTH>(p, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------
// E for the vertex-colour mode: gradient w.r.t. the per-vertex colours, fill-back by index
// ---------------------------------------------------------------------------------------
struct GatherVCParams {
    const float* verts;     // [B,V,3] projected
    const int32_t* fidx;    // [B,F0,3]
    const int32_t* fim;     // raster orientation, values in [0, 2 F0)
    const float* grad_rgb;  // image orientation
    float* grad_vcolors;    // [B,V,3], pre-zeroed, accumulated with fp32 atomics
    int B, V, F0, fill_back, is;
    float eps;
    int dbg;    // profiling experiments (flags >> 8)
    int texel;  // texel layout code of the vertex-colour texture (mr_common.hpp: texel_vertex)
};

// contribution of one won pixel to the colours of the face's three vertices (its own order)
__device__ __forceinline__ void gather_vc_pixel(const GatherVCParams& p, const Face& f, int b, int xi, int yi,
                                                float (*acc)[3]) {
    float w[3], zp, tif[3], g[3];
    bary(f, xi, yi, zp, w);
    tex_coords(w, zp, f.v, 2, p.eps, tif);
#pragma unroll
    for (int c = 0; c < 3; c++) g[c] = p.grad_rgb[idx3<true>(b, yi, xi, c, p.is)];
    // taps pn = 1, 2, 4 are the texels holding the colours of vertices 0, 1, 2
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int pn = 1 << k;
        float wg = 1.0f;
#pragma unroll
        for (int j = 0; j < 3; j++) wg *= ((pn >> j) & 1) ? (tif[j] - 0.0f) : (1.0f - (tif[j] - 0.0f));
#pragma unroll
        for (int c = 0; c < 3; c++) acc[k][c] += wg * g[c];
    }
}

// GLPF lanes per REAL face, 16 faces per wave.  Pass 0 handles every face's visible orientation
// (normally exactly one of the two fill-back orientations is front-facing), pass 1 the second
// orientation of zero-area faces.  Per pass, three lock-step stages mirror the forward kernel:
//   P1 one lane of the quad: orientation's vertices + inverse into an LDS face cache;
//   P2 the quad probes the face's bbox in face_index_map (lane k: rows k, k+4, ...; 8 probes in
//      flight per iteration) and appends the pixels the face WON as fragments (slot, dx, dy) to
//      an LDS ring (ballot compaction); faces with a large bbox are probed by the whole wave;
//   P3 lane per fragment, 64 at a time: barycentrics, sampling weights, gradient of the three
//      vertex colours, accumulated per face in LDS (ds_add_f32).
// The per-face sums go to grad_vcolors with 9 global fp32 atomics per live face.
constexpr int GV_SLOTS = MR_WAVE / GLPF;  // faces per wave
constexpr int GV_FCS = 23;    // dwords per face-cache slot (odd stride)
constexpr int GV_FQ = 1024;   // fragment ring capacity (>= 63 + 8 * 64, power of 2)

template <int DUMMY>
__global__ void __launch_bounds__(256) gather_vc_kernel(GatherVCParams p) {
    __shared__ float fcache[4][GV_SLOTS * GV_FCS];
    __shared__ float facc[4][GV_SLOTS * 9];
    __shared__ unsigned fragq[4][GV_FQ];

    const int64_t total = (int64_t)p.B * p.F0;
    const int64_t gid = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    const int64_t i = gid / GLPF;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % GLPF, slot_own = lane / GLPF;
    const bool valid = i < total;
    const int b = valid ? (int)(i / p.F0) : 0;
    const int f0 = valid ? (int)(i % p.F0) : 0;
    const int is = p.is;
    float* fc = fcache[wave];
    float* fa = facc[wave];
    unsigned* fq = fragq[wave];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    int vid[3] = {0, 0, 0};
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = __builtin_nanf("");
    if (valid) {
        const int32_t* ix = p.fidx + i * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            vid[k] = ix[k];
            const float* g = p.verts + ((int64_t)b * p.V + vid[k]) * 3;
            v[3 * k] = g[0]; v[3 * k + 1] = g[1]; v[3 * k + 2] = g[2];
        }
    }
    float acc[3][3];  // [real vertex][channel]
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) acc[k][c] = 0.0f;
    const int32_t* fim_b = p.fim + (int64_t)b * is * is;
    if (p.dbg & 8) {
        if (valid && v[0] == 12345.0f) p.grad_vcolors[0] = v[1] + v[5];
        return;
    }

    // P3: one lane per fragment
    int qhead = 0, qn = 0;
    auto shade = [&](int n) {
        if (lane < n && !(p.dbg & 2)) {
            const unsigned fr = fq[(qhead + lane) & (GV_FQ - 1)];
            const int slot = (int)(fr >> 26);
            const float* c = fc + slot * GV_FCS;
            Face f;
#pragma unroll
            for (int k = 0; k < 9; k++) f.inv[k] = c[9 + k];
            f.v[2] = c[2]; f.v[5] = c[5]; f.v[8] = c[8];
            const int xi = __float_as_int(c[18]) + (int)(fr & 0x1fffu), yi = __float_as_int(c[19]) + (int)((fr >> 13) & 0x1fffu);
            const int bb = __float_as_int(c[20]);
            float w[3], zp, tif[3], g[3];
            bary(f, xi, yi, zp, w);
            tex_coords(w, zp, f.v, 2, p.eps, tif);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) g[ch] = p.grad_rgb[idx3<true>(bb, yi, xi, ch, is)];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int pn = 1 << k;
                float wg = 1.0f;
#pragma unroll
                for (int j = 0; j < 3; j++) wg *= ((pn >> j) & 1) ? (tif[j] - 0.0f) : (1.0f - (tif[j] - 0.0f));
#pragma unroll
                for (int ch = 0; ch < 3; ch++) atomicAdd(&fa[slot * 9 + k * 3 + ch], wg * g[ch]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // which orientations can be visible (normally exactly one: the front-facing one)
    bool live[2];
    {
        float r[9];
#pragma unroll
        for (int k = 0; k < 3; k++) { r[k] = v[6 + k]; r[3 + k] = v[3 + k]; r[6 + k] = v[k]; }
        const FaceBox ba = face_box(v, is), bb2 = face_box(r, is);
        live[0] = valid && ba.x0 <= ba.x1;
        live[1] = valid && p.fill_back && bb2.x0 <= bb2.x1;
    }
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        // pass 0: every face's first live orientation; pass 1: the second one where BOTH are live
        // (zero-area faces only)
        const int o = pass == 0 ? (live[0] ? 0 : 1) : 1;
        const bool mine = pass == 0 ? (live[0] || live[1]) : (live[0] && live[1]);
        Face f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int s = o ? 2 - k : k;
            f.v[3 * k] = v[3 * s]; f.v[3 * k + 1] = v[3 * s + 1]; f.v[3 * k + 2] = v[3 * s + 2];
        }
        FaceBox bx = face_box(f.v, is);
        if (!mine) { bx.x0 = 1; bx.x1 = 0; bx.y0 = 1; bx.y1 = 0; }
        const bool nonempty = mine && bx.x0 <= bx.x1;
        const int bw = bx.x1 - bx.x0 + 1, bh = bx.y1 - bx.y0 + 1;
        const bool big = nonempty && bw * bh > GATHER_BIG;
        const int fn = o ? f0 + p.F0 : f0;
        if (__ballot(nonempty) == 0ull) continue;  // wave-uniform
        if (p.dbg & 16) {
            if (nonempty && bw == 12345) p.grad_vcolors[0] = 1.0f;
            continue;
        }

        // P1: park the orientation's face in the cache, clear its accumulators
        if (sub == 0) {
            float* c = fc + slot_own * GV_FCS;
            if (nonempty) {
                face_inverse(f.v, f.inv, is);
#pragma unroll
                for (int k = 0; k < 9; k++) { c[k] = f.v[k]; c[9 + k] = f.inv[k]; }
                c[18] = __int_as_float((int)bx.x0); c[19] = __int_as_float((int)bx.y0); c[20] = __int_as_float(b);
            }
#pragma unroll
            for (int k = 0; k < 9; k++) fa[slot_own * 9 + k] = 0.0f;
        }
        __builtin_amdgcn_wave_barrier();

        // P2 (large faces): the whole wave probes the bbox, lane per column, rows in groups of 8;
        // won pixels join the same fragment ring
        unsigned long long m_big = (p.dbg & 4) ? 0ull : __ballot(big && sub == 0);
        while (m_big) {
            const int src = __ffsll((long long)m_big) - 1;
            m_big &= m_big - 1;
            const int x0 = __shfl((int)bx.x0, src), y0 = __shfl((int)bx.y0, src);
            const int w_ = __shfl(bw, src), h_ = __shfl(bh, src);
            const int ff = __shfl(fn, src);
            const int32_t* fim_s = p.fim + (int64_t)__shfl(b, src) * is * is;  // the wave may straddle two images
            for (int cx0 = 0; cx0 < w_; cx0 += MR_WAVE) {
                const int cx = cx0 + lane;
                const bool col = cx < w_;
                for (int r = 0; r < h_; r += 8) {
                    int hit[8];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        hit[u] = col ? fim_s[min(y0 + r + u, y0 + h_ - 1) * is + x0 + cx] : -2;
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const bool won = col && r + u < h_ && hit[u] == ff;
                        const unsigned long long m = __ballot(won);
                        if (won)
                            fq[(qhead + qn + __popcll(m & lt_mask)) & (GV_FQ - 1)] =
                                ((unsigned)(src / GLPF) << 26) | ((unsigned)(r + u) << 13) | (unsigned)cx;
                        qn += __popcll(m);
                    }
                    __builtin_amdgcn_wave_barrier();
                    while (qn >= MR_WAVE) {
                        shade(MR_WAVE);
                        qhead = (qhead + MR_WAVE) & (GV_FQ - 1);
                        qn -= MR_WAVE;
                    }
                }
            }
        }

        // P2: GLPF lanes per face (rows sub, sub + GLPF, ...), 8 probes per iteration, lock step
        bool act = nonempty && !big && !(p.dbg & 4);
        int px = bx.x0, py = bx.y0 + sub;
        act = act && py <= bx.y1;
        while (__ballot(act) != 0ull) {
            int hit[8];
#pragma unroll
            for (int u = 0; u < 8; u++) hit[u] = act ? fim_b[py * is + min(px + u, (int)bx.x1)] : -2;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool won = act && px + u <= bx.x1 && hit[u] == fn;
                const unsigned long long m = __ballot(won);
                if (won)
                    fq[(qhead + qn + __popcll(m & lt_mask)) & (GV_FQ - 1)] =
                        ((unsigned)slot_own << 26) | ((unsigned)(py - bx.y0) << 13) | (unsigned)(px + u - bx.x0);
                qn += __popcll(m);
            }
            if (act) {
                px += 8;
                if (px > bx.x1) { px = bx.x0; py += GLPF; }
                act = py <= bx.y1;
            }
            __builtin_amdgcn_wave_barrier();
            while (qn >= MR_WAVE) {
                shade(MR_WAVE);
                qhead = (qhead + MR_WAVE) & (GV_FQ - 1);
                qn -= MR_WAVE;
            }
        }
        if (qn > 0) {
            shade(qn);
            qhead = (qhead + qn) & (GV_FQ - 1);
            qn = 0;
        }
        // collect this orientation's sums, mapped back to the real vertex order
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                // tap k of this orientation holds the colour of its vertex texel_vertex(k), i.e. of the REAL face's
                // vertex at position (o ? 2 - that : that)
                const int tv = texel_vertex(p.texel, k, o != 0);
                const int real = o ? 2 - tv : tv;
                const float add = fa[slot_own * 9 + k * 3 + c];
#pragma unroll
                for (int q = 0; q < 3; q++) acc[q][c] += (q == real) ? add : 0.0f;
            }
        __builtin_amdgcn_wave_barrier();
    }
    if (valid && sub == 0 && !(p.dbg & 1)) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float* o = p.grad_vcolors + ((int64_t)b * p.V + vid[k]) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++)
                if (acc[k][c] != 0.0f) atomicAdd(&o[c], acc[k][c]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// E for the vertex-colour mode, pixel-parallel: the default when the per-image colour table fits LDS
// ---------------------------------------------------------------------------------------
// One block owns a SV_W x SV_H pixel region of one image, one lane per pixel (SV_RPW rows per
// wave).  face_index_map is read once, coalesced (a wave = one 64-pixel row segment, 256 B);
// regions without geometry leave after that single load.  A covered pixel needs its rgb gradient,
// the winner's index triple and -- STORED: the barycentric weights and depth the forward pass
// wrote for this very pixel plus the three vertex depths; else: the three projected vertices, from
// which inverse, weights and depth are recomputed with the forward's arithmetic.  All of these
// loads are issued before the first use (two round trips after face_index_map).  The 9 products
// (sampling weight x gradient channel) are added into a block-private [V,3] table in LDS, which is
// flushed with one global fp32 atomic per touched (vertex, channel): a few thousand per image
// instead of 9 per visible face, and no probing of face bounding boxes at all.
//
// The LDS table is 64-bit FIXED POINT, not fp32: ds_add_f32 is several times slower than the
// integer LDS atomics on gfx950 (measured: the 9 float adds per pixel cost more than everything
// else in this kernel together).  Every product w * g is bounded by the largest |g| of the region
// (the sampling weights are in [0,1]), so with that maximum < 2^e the products are scaled by
// 2^(SV_FIX_BITS - e), truncated to int64 and summed exactly; <= 3 * SV_W * SV_H terms of
// magnitude < 2^SV_FIX_BITS cannot overflow.  The absolute error per term is 2^-50 of the region's
// largest gradient -- far below fp32 rounding of the sum -- and the block's sums do not depend on
// the order of the additions.  Regions whose gradient holds an Inf / NaN take a plain fp32
// global-atomic path so that non-finite values propagate as they do in the gather kernel.
constexpr int SV_WAVES = 16;                                    // waves per block
constexpr int SV_RPW = 2, SV_W = 64, SV_H = SV_WAVES * SV_RPW;  // rows per wave, region
constexpr int SV_MAX_TABLE_BYTES = 60 * 1024;
constexpr int SV_FIX_BITS = 50;
static_assert(3LL * SV_W * SV_H < (1LL << (63 - SV_FIX_BITS)), "fixed-point sums must not overflow");

struct ScatterVCParams {
    GatherVCParams g;
    const float* weight;  // STORED: [B,is,is,3] raster orientation
    const float* depth;   // STORED: [B,is,is] image orientation
    int rx_n, ry_n;
};

template <bool STORED>
__global__ void __launch_bounds__(SV_WAVES * MR_WAVE) scatter_vc_kernel(ScatterVCParams sp) {
    extern __shared__ long long vtab[];  // [V * 3] rounded up to an even count
    __shared__ unsigned wmax[SV_WAVES];

    const GatherVCParams& p = sp.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned blk = xcd_remap(blockIdx.x, gridDim.x);
    const int rx = (int)(blk % (unsigned)sp.rx_n), ry = (int)((blk / (unsigned)sp.rx_n) % (unsigned)sp.ry_n);
    const int b = (int)(blk / (unsigned)(sp.rx_n * sp.ry_n));
    const int is = p.is;
    const int32_t* fim_b = p.fim + (int64_t)b * is * is;
    const int xi = rx * SV_W + lane, yi0 = ry * SV_H + wave * SV_RPW;

    int fnv[SV_RPW];
#pragma unroll
    for (int r = 0; r < SV_RPW; r++) fnv[r] = (xi < is && yi0 + r < is) ? fim_b[(yi0 + r) * is + xi] : -1;
    bool cov = false;
#pragma unroll
    for (int r = 0; r < SV_RPW; r++) cov = cov || fnv[r] >= 0;
    if (!__syncthreads_or(cov) || (p.dbg & 8)) return;  // block-uniform

    const float* verts_b = p.verts + (int64_t)b * p.V * 3;
    const int32_t* fidx_b = p.fidx + (int64_t)b * p.F0 * 3;
    float g[SV_RPW][3], w[SV_RPW][3], zp[SV_RPW];
    int vid[SV_RPW][3];
    bool rev_r[SV_RPW];
#pragma unroll
    for (int r = 0; r < SV_RPW; r++) {
        const bool won = fnv[r] >= 0;
        const bool o = fnv[r] >= p.F0;  // reversed copy of face fn - F0
        rev_r[r] = o;
        const int32_t* ix = fidx_b + (int64_t)(won ? (o ? fnv[r] - p.F0 : fnv[r]) : 0) * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) vid[r][k] = won ? ix[o ? 2 - k : k] : 0;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) g[r][ch] = won ? p.grad_rgb[idx3<true>(b, yi0 + r, xi, ch, is)] : 0.0f;
        if (STORED) {
            const float* wq = sp.weight + idx1<false>(b, yi0 + r, xi, is) * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) w[r][k] = won ? wq[k] : 0.0f;
            zp[r] = won ? sp.depth[idx1<true>(b, yi0 + r, xi, is)] : 1.0f;
        }
    }
    Face f[SV_RPW];  // STORED: only the three depths v[2], v[5], v[8] are loaded
#pragma unroll
    for (int r = 0; r < SV_RPW; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float* q = verts_b + (int64_t)vid[r][k] * 3;
            const bool won = fnv[r] >= 0;
            if (!STORED) { f[r].v[3 * k] = won ? q[0] : 0.0f; f[r].v[3 * k + 1] = won ? q[1] : 0.0f; }
            f[r].v[3 * k + 2] = won ? q[2] : 1.0f;
        }

    // largest |gradient| over the region's covered pixels, as float bits
    unsigned mx = 0u;
#pragma unroll
    for (int r = 0; r < SV_RPW; r++)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) mx = max(mx, __float_as_uint(g[r][ch]) & 0x7fffffffu);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
    if (lane == 0) wmax[wave] = mx;
    const int n2 = (p.V * 3 + 1) >> 1;
    for (int k = threadIdx.x; k < n2; k += blockDim.x) reinterpret_cast<int4*>(vtab)[k] = make_int4(0, 0, 0, 0);
    if (p.dbg & 16) {
        if (f[0].v[2] + f[SV_RPW - 1].v[8] == 12345.0f) p.grad_vcolors[0] = 1.0f;
        return;
    }
    __syncthreads();  // table zeroed, maxima visible

    unsigned bm = 0u;
#pragma unroll
    for (int k = 0; k < SV_WAVES; k++) bm = max(bm, wmax[k]);
    if (bm == 0u) return;                   // every gradient of the region is +-0: nothing to add
    const bool finite = bm < 0x7f800000u;   // else: fp32 global atomics, Inf / NaN propagate
    const int shift = SV_FIX_BITS - ((int)(bm >> 23) - 126);  // largest |g| < 2^((bm >> 23) - 126)
    float* out = p.grad_vcolors + (int64_t)b * p.V * 3;

#pragma unroll
    for (int r = 0; r < SV_RPW; r++) {
        if (fnv[r] >= 0 && !(p.dbg & 2)) {
            float tif[3];
            if (!STORED) {
                face_inverse(f[r].v, f[r].inv, is);
                bary(f[r], xi, yi0 + r, zp[r], w[r]);
            }
            tex_coords(w[r], zp[r], f[r].v, 2, p.eps, tif);
            // taps pn = 1, 2, 4 are the texels holding the colours of vertices 0, 1, 2
            float val[9];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int pn = 1 << k;
                float wg = 1.0f;
#pragma unroll
                for (int j = 0; j < 3; j++) wg *= ((pn >> j) & 1) ? (tif[j] - 0.0f) : (1.0f - (tif[j] - 0.0f));
#pragma unroll
                for (int ch = 0; ch < 3; ch++) val[k * 3 + ch] = wg * g[r][ch];
            }
            // Neighbouring lanes hold pixels of the same face or of faces sharing a vertex, and an
            // LDS atomic serialises lanes that hit one address.  Lane l therefore issues its nine
            // adds in the order (l % 9), (l % 9) + 1, ... so that the lanes of a run work on
            // different (vertex, channel) cells in any one instruction.
            const int rot = lane % 9;
#pragma unroll
            for (int bit = 1; bit < 16; bit <<= 1) {
                float t[9];
#pragma unroll
                for (int k = 0; k < 9; k++) t[k] = val[(k + bit) % 9];
                const bool on = (rot & bit) != 0;
#pragma unroll
                for (int k = 0; k < 9; k++) val[k] = on ? t[k] : val[k];
            }
#pragma unroll
            for (int j = 0; j < 9; j++) {
                int sl = j + rot;
                sl = sl >= 9 ? sl - 9 : sl;
                const int k = (sl >= 3) + (sl >= 6), ch = sl - 3 * k;
                const int cell = sel3(vid[r][0], vid[r][1], vid[r][2], texel_vertex(p.texel, k, rev_r[r])) * 3 + ch;
                if (p.dbg & 4) {
                    if (val[j] == 12345.0f) vtab[0] = 1;
                } else if (finite) {
                    const long long q = (long long)ldexp((double)val[j], shift);
                    atomicAdd(reinterpret_cast<unsigned long long*>(&vtab[cell]), (unsigned long long)q);
                } else if (val[j] != 0.0f) {
                    atomicAdd(&out[cell], val[j]);
                }
            }
        }
    }
    if (!finite || (p.dbg & 1)) return;  // block-uniform
    __syncthreads();
    for (int k = threadIdx.x; k < p.V * 3; k += blockDim.x) {
        const long long t = vtab[k];
        if (t != 0) {
            const float v = (float)ldexp((double)t, -shift);
            if (v != 0.0f) atomicAdd(&out[k], v);
        }
    }
}

// ---------------------------------------------------------------------------------------
// E for the flow-mode render (mr_render_flow_backward): tile-persistent scatter
// ---------------------------------------------------------------------------------------
// The forward pass left one byte per (32x8 tile, row pair) saying whether anything is covered there (~17 % of the
// tiles of a hand + object frame).  ST_G workgroups of 4 waves share an image; every wave walks its own interleaved
// subset of the image's tiles, skips the empty ones on that byte (one scalar load -- the region kernel above
// dispatches 16 waves per 64x32 region only to find 70 % of them empty: 16 us of its 50 are wave dispatch), and
// handles a covered tile alone: lane = 4 consecutive pixels of a row, so that one wave-wide 16-byte load fetches a
// whole tile of a plane and a lane has four independent gather chains (face -> vertex ids -> vertex depths) in
// flight.  The [V,3] fixed-point table in LDS is zeroed and flushed once per WORKGROUP (a quarter image), not once
// per region; its scale comes from a first pass over the gradient of the workgroup's covered tiles.
// FLOWGRAD: the incoming gradient is the FLOW-space gradient [B,H,W,2] plus the masks of opticalflow.py:146-154 --
// the adjoint of crop / permute / mask products is applied on the fly with the arithmetic of
// flow_finalize_backward_kernel ((g * (m_x * occl)) * m_pre; third colour plane: zero), so that the [B,3,is,is]
// colour-space gradient is never materialised.
// Launch shape.  The [V, channels] fixed-point table is what limits residency (42.7 KB for a hand + object mesh with
// three channels: 3 workgroups per compute unit, 768 for 2048 launched, i.e. three rounds of ~9 us each -- workgroup
// timeline, scripts/bwd_timeline.py).  The flow-space gradient has two channels (the third colour plane's gradient is
// identically zero): its table is a third smaller; and eight workgroups of EIGHT waves per image instead of sixteen of
// four keep the 64 waves per image while halving the per-image fixed work (covered-tile list, table zeroing, flush):
// 4 x 8 waves fit a compute unit, so all 1024 workgroups of a 128-image launch are resident at once.
constexpr int ST_G = 8;       // workgroups per image (sp.groups; profiling: flags >> 8 bits 4-6 select 2 / 4 / 16 / 32)
constexpr int ST_WAVES = 8;   // waves per workgroup
constexpr int ST_TW = 32, ST_TH = 8;  // the forward's tile
constexpr int ST_MAX_TILES = 4096;    // tiles per image the covered-tile list in LDS can hold (1024 x 1024 pixels)
// fixed point of the per-workgroup sums: terms below 2^ST_FIX_BITS in magnitude (2^-37 of the workgroup's largest gradient
// per term: 13 bits below the last bit of an fp32 value of that size), sums below 2^ST_SUM_BITS (see scatter_tiles_body)
constexpr int ST_FIX_BITS = 37, ST_SUM_BITS = 50;

struct ScatterTilesParams {
    GatherVCParams g;
    const float* weight;        // [B,is,is,3] raster orientation, valid at covered pixels
    const float* depth;         // [B,is,is] image orientation, valid at covered pixels
    const uint32_t* tile_hit;   // [B, tiles] one byte per wave (row pair) of the forward's tile kernel; nullable
    const int32_t* vid_map;     // nullable: [B,is,is,3] vertex ids of the winner, written by mr_render_flow_forward; then
                                // `weight` holds the three sampling weights of the colour taps (not barycentrics)
    // FLOWGRAD
    const float* grad_flow;     // [B,H,W,2]
    const float* m_pre;         // [B,is,is] image orientation
    const float* m_x_lo;        // images [0, split)
    const float* m_x_hi;        // images [split, B)
    const float* occl;
    int split, H, W;
    int tiles_x, tiles_y;
    int groups;  // workgroups per image
    const float* grad_bound;  // nullable (FLOWGRAD): [B] upper bounds of |grad_flow| per image
    // PAIR (mr_flow_pair_backward_tiles): the pair loss's backward folded in -- pass 1 computes the flow gradient of the
    // workgroup's tiles itself (it used to be a kernel of its own + 8 B per pixel written and read back)
    float* stash;             // [B,H,W,2] scratch: the masked flow-space gradient, pass 1 -> pass 2 of the same workgroup
    const float* flow;        // [B,H,W,2] the stacked final flows (images [0, split): flow12, [split, B): flow21)
    const float* image_ref;   // [split,3,H,W]
    const float* image;
    const float* jitter_ref;  // [split,Cj,H,W]
    const float* jitter;
    int Cj;
    const float* sums;        // [split,4] of the pair loss's forward
    const float* gl_fwd;      // [split] incoming gradients of loss_fwd / loss_bwd (gl_bwd nullable)
    const float* gl_bwd;
    // UNIT (mr_pair_step_backward): two more incoming gradients, added to both directions' coefficients: of loss_bwd + loss_fwd
    // [split] and of the mean over the batch [1] (a sample's share = *gl_mean / mean_div), each nullable
    const float* gl_sum;
    const float* gl_mean;
    float mean_div;
    int mean_of;              // 1: the mean is over loss_fwd only -- no share for the other direction
    float pair_thresh;
    // UNIT (mr_flow_pair_backward_unit_tiles): the forward (mr_flow_pair_forward_grad_tiles) left the pair loss's gradient for
    // a coefficient of 1, masks applied; this launch multiplies by grad_loss / count of the image -- no taps, no pass 1
    const float* unit_grad;   // [B,H,W,2]
    const float* unit_max;    // [B] largest |unit gradient| per image (an upper bound is enough)
    // WORK (round 5): the images' covered-tile lists the forward's finalize launch left (mr_common.hpp, ScatterWork)
    ScatterWork work;
};

template <bool FLOWGRAD, bool PAIR = false, bool UNIT = false>
__device__ __forceinline__ void st_load_grad(const ScatterTilesParams& sp, int b, int yi, int x, float (*g)[3], float ucoef = 0.0f) {
    const GatherVCParams& p = sp.g;
    const int is = p.is;
    const int yimg = is - 1 - yi;
    if (PAIR || UNIT) {  // pass 1 of this workgroup (UNIT: the forward launch) left the gradient, masks applied
        const float* st = UNIT ? sp.unit_grad : sp.stash;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float2 gf = make_float2(0.0f, 0.0f);
            if (yimg < sp.H && x + j < sp.W) gf = *reinterpret_cast<const float2*>(st + (((int64_t)b * sp.H + yimg) * sp.W + x + j) * 2);
            g[j][0] = UNIT ? gf.x * ucoef : gf.x; g[j][1] = UNIT ? gf.y * ucoef : gf.y; g[j][2] = 0.0f;
        }
    } else if (!FLOWGRAD) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float4 v = *reinterpret_cast<const float4*>(p.grad_rgb + (((int64_t)b * 3 + ch) * is + yimg) * is + x);
            g[0][ch] = v.x; g[1][ch] = v.y; g[2][ch] = v.z; g[3][ch] = v.w;
        }
    } else {
        const int64_t o = ((int64_t)b * is + yimg) * is + x;
        const float* mx = b < sp.split ? sp.m_x_lo + o : sp.m_x_hi + (o - (int64_t)sp.split * is * is);
        const float4 a4 = *reinterpret_cast<const float4*>(sp.m_pre + o);
        const float4 x4 = *reinterpret_cast<const float4*>(mx);
        const float4 o4 = *reinterpret_cast<const float4*>(sp.occl + o);
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
        const float post[4] = {x4.x * o4.x, x4.y * o4.y, x4.z * o4.z, x4.w * o4.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float2 gf = make_float2(0.0f, 0.0f);
            const bool in = yimg < sp.H && x + j < sp.W;
            if (in) gf = *reinterpret_cast<const float2*>(sp.grad_flow + (((int64_t)b * sp.H + yimg) * sp.W + x + j) * 2);
            g[j][0] = in ? (gf.x * post[j]) * a[j] : 0.0f;
            g[j][1] = in ? (gf.y * post[j]) * a[j] : 0.0f;
            g[j][2] = 0.0f;
        }
    }
}

#ifdef MR_WG_TIMELINE
__device__ unsigned long long mr_dbg_st[4096 * 8];  // profiling builds: phase stamps of the first 4096 workgroups
#define MR_ST_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) mr_dbg_st[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define MR_ST_STAMP(k) do { } while (0)
#endif

template <bool FLOWGRAD, bool REC, bool PAIR, bool UNIT = false, bool WORK = false>
__device__ __forceinline__ void scatter_tiles_body(const ScatterTilesParams& sp) {
    MR_ST_STAMP(0);
    extern __shared__ long long vtab[];  // [V * NCH] rounded up to an even count
    constexpr int NCH = FLOWGRAD ? 2 : 3;  // (the flow-space gradient has no third channel)
    __shared__ unsigned wmax[ST_WAVES];
    __shared__ unsigned short hits[WORK ? 1 : ST_MAX_TILES];  // the image's covered tiles, ascending
    const GatherVCParams& p = sp.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int G = sp.groups;
    // PAIR: an image's workgroups on the XCD whose L2 holds what the forward launches just wrote for it (their tile lists are
    // image-major and every XCD takes a contiguous eighth: images [x B / 8, (x + 1) B / 8) on XCD x, more or less)
    const unsigned lidx = (PAIR || FLOWGRAD) ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    int b = (int)(lidx / (unsigned)G), part = (int)(lidx % (unsigned)G);
    const int is = p.is;
    const int T = sp.tiles_x * sp.tiles_y;
    int n_hits = 0;
    // WORK: the table is zeroed FIRST (it needs nothing but V), so that the barrier behind it can wait where the first LDS
    // atomic is about to be issued -- behind the tile's loads instead of in front of them
    if constexpr (WORK)
        for (int k = threadIdx.x; k < (p.V * NCH + 1) >> 1; k += blockDim.x) reinterpret_cast<int4*>(vtab)[k] = make_int4(0, 0, 0, 0);
    const unsigned short* cov_b = nullptr;  // WORK: this workgroup's share of the forward's list of the image's covered tiles
    int t_first = 0;
    auto tile_at = [&](int h) -> int { return WORK ? (h == wave ? t_first : (int)cov_b[h]) : (int)hits[h]; };
    if constexpr (WORK) {
        // WORK (round 5): the launch's workgroups are handed out over the images IN PROPORTION TO THEIR COVERED TILES.  With a
        // fixed number per image the images that cover more than G x ST_WAVES tiles need a second round of their waves while
        // the waves of the others idle: on the bench scene (45 covered tiles per image on average, 8 x 8 waves) a third of the
        // workgroups ran two rounds and the launch ended at 18.8 us with the average workgroup done at 12.5
        // (profiles/r05_bwd_timeline_before.txt).  Every wave derives the split from the per-image counts the forward's
        // finalize launch left: q = tiles per workgroup (at least one per wave) such that sum ceil(n_b / q) fits the grid,
        // image b gets ceil(n_b / q) workgroups sharing its n_b tiles evenly, and the workgroups in use are spread over the
        // logical index range so that every XCD gets the same share.  No listing pass, no LDS list.
        const int B2 = p.B, grid = (int)gridDim.x;
        const int* ncov = sp.work.n_cov;
        // (every quantity below is the same in all lanes: readlane keeps it in scalar registers; the scans are DPP row shifts,
        // six vector instructions each, no LDS crossbar)
        // First try: one tile per wave (q = ST_WAVES) -- it fits whenever the scene leaves the grid some room, and then ONE pass
        // over the counts gives everything; else q = ceil(N / room) and a second pass.  The counts of the first 4 x 64 images
        // are requested TOGETHER and stay in registers for all passes (one round trip; further images: read again, from L1).
        constexpr int NC = 4;
        int nreg[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) nreg[i] = i * MR_WAVE + lane < B2 ? ncov[i * MR_WAVE + lane] : 0;
        int q = ST_WAVES, used = 0, N = 0;
#pragma unroll
        for (int i = 0; i < NC; i++) {
            if (i * MR_WAVE >= B2) break;
            N += __builtin_amdgcn_readlane(wave_incl_sum(nreg[i], lane), MR_WAVE - 1);
            used += __builtin_amdgcn_readlane(wave_incl_sum((nreg[i] + ST_WAVES - 1) / ST_WAVES, lane), MR_WAVE - 1);
        }
        for (int c = NC * MR_WAVE; c < B2; c += MR_WAVE) {
            const int n = c + lane < B2 ? ncov[c + lane] : 0;
            N += __builtin_amdgcn_readlane(wave_incl_sum(n, lane), MR_WAVE - 1);
            used += __builtin_amdgcn_readlane(wave_incl_sum((n + ST_WAVES - 1) / ST_WAVES, lane), MR_WAVE - 1);
        }
        if (used > grid) {
            const int room = max(grid - B2, 1);  // (sum ceil(n_b / q) <= N / q + B2)
            q = max(ST_WAVES, (N + room - 1) / room);
            used = 0;
#pragma unroll
            for (int i = 0; i < NC; i++) {
                if (i * MR_WAVE >= B2) break;
                used += __builtin_amdgcn_readlane(wave_incl_sum((nreg[i] + q - 1) / q, lane), MR_WAVE - 1);
            }
            for (int c = NC * MR_WAVE; c < B2; c += MR_WAVE) {
                const int n = c + lane < B2 ? ncov[c + lane] : 0;
                used += __builtin_amdgcn_readlane(wave_incl_sum((n + q - 1) / q, lane), MR_WAVE - 1);
            }
            if (used > grid) used = grid;  // (grid <= images: cannot be; the surplus images would go without a gradient)
        }
        // logical index -> slot k of the `used` workgroups with work.  xcd_remap gave XCD x the logical indices [x grid / 8,
        // (x + 1) grid / 8): it takes the slots [x used / 8, (x + 1) used / 8) with the first of them -- every XCD the same share
        // of the work, images in list order (as the forward's tile list hands them to the XCDs), divisions by 8 only
        int k;
        if ((grid & 7) == 0) {
            const int per = grid >> 3, x = (int)lidx / per, j = (int)lidx - x * per;
            const int k0 = (int)(((long long)x * used) >> 3), k1 = (int)(((long long)(x + 1) * used) >> 3);
            if (j >= k1 - k0) return;
            k = k0 + j;
        } else {
            if ((int)lidx >= used) return;
            k = (int)lidx;
        }
        int base = 0, nb = 0, pb = 1;
        b = -1;
        for (int c = 0; c < B2 && b < 0; c += MR_WAVE) {
            const int ci = c / MR_WAVE;
            const int n = ci == 0 ? nreg[0] : ci == 1 ? nreg[1] : ci == 2 ? nreg[2] : ci == 3 ? nreg[3]
                                                                                         : (c + lane < B2 ? ncov[c + lane] : 0);
            const int parts = (n + q - 1) / q;
            const int incl = wave_incl_sum(parts, lane);
            const unsigned long long mine = __ballot(k >= base + incl - parts && k < base + incl);
            if (mine != 0ull) {
                const int src = __ffsll((long long)mine) - 1;
                b = c + src;
                part = k - (base + __builtin_amdgcn_readlane(incl - parts, src));
                nb = __builtin_amdgcn_readlane(n, src);
                pb = __builtin_amdgcn_readlane(parts, src);
            }
            base += __builtin_amdgcn_readlane(incl, MR_WAVE - 1);
        }
        if (b < 0) return;  // (cannot happen: k < used)
        // this workgroup's share of the image's list: tiles [first, first + n_hits); its waves take them round-robin
        const int each = nb / pb, rem = nb % pb;
        const int first = part * each + min(part, rem);
        n_hits = each + (part < rem ? 1 : 0);
        cov_b = sp.work.cov + (int64_t)b * T + first;
        part = 0; G = 1;
        // (the wave's first tile id is requested now: its round trip runs beside the image's scalars and the table zeroing)
        t_first = wave < n_hits ? (int)cov_b[wave] : 0;
    }
    const int r = lane >> 3, x4 = (lane & 7) * 4;  // the lane's pixel quad inside a tile
    const int32_t* fim_b = p.fim + (int64_t)b * is * is;
    const float* verts_b = p.verts + (int64_t)b * p.V * 3;
    const int32_t* fidx_b = p.fidx + (int64_t)b * p.F0 * 3;

    // UNIT: the image's scalars (count, incoming loss gradient, bound of the unit gradient) are requested NOW, in front of
    // the coverage words, not behind the list they do not depend on (one cold round trip less on the workgroup's chain)
    float u_cnt = 1.0f, u_gl = 0.0f, u_max = 0.0f;
    if constexpr (UNIT) {
        const int dir = b >= sp.split ? 1 : 0, pb = b - dir * sp.split;
        const float* gl = dir ? sp.gl_fwd : sp.gl_bwd;
        u_cnt = sp.sums[pb * 4 + (dir ? 1 : 3)];
        u_gl = gl ? gl[pb] : 0.0f;
        if (sp.gl_sum) u_gl += sp.gl_sum[pb];
        if (sp.gl_mean && (dir || !sp.mean_of)) u_gl += sp.gl_mean[0] / sp.mean_div;
        u_max = sp.unit_max[b];
    }
    // the list of covered tiles (block-wide compaction of the coverage bytes); the waves that share an image then take
    // them round-robin: every wave gets its share whatever part of the screen the mesh sits in.  (All coverage words
    // requested up front and compacted behind one barrier -- eight predicated rounds unrolled -- measured SLOWER than
    // this loop at 1024 tiles per image: 1.98 against 1.54 us.)
    __shared__ int wcnt[ST_WAVES];
    for (int t0 = 0; t0 < T && !WORK; t0 += blockDim.x) {
        const int t = t0 + threadIdx.x;
        const bool hit = t < T && (sp.tile_hit ? sp.tile_hit[(int64_t)b * T + t] != 0u : true);
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int base = n_hits;
        for (int w = 0; w < wave; w++) base += wcnt[w];
        if (hit) hits[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)t;
        for (int w = 0; w < ST_WAVES; w++) n_hits += wcnt[w];
        __syncthreads();
    }

    MR_ST_STAMP(1);
    if (part * ST_WAVES >= n_hits) return;  // fewer covered tiles than waves before this workgroup: nothing to do
    const int n2 = (p.V * NCH + 1) >> 1;
    for (int k = threadIdx.x; k < n2 && !WORK; k += blockDim.x) reinterpret_cast<int4*>(vtab)[k] = make_int4(0, 0, 0, 0);

    // pass 1: largest |gradient| over the workgroup's covered tiles, as float bits -- or the caller's bound for the image
    unsigned mx = 0u;
    const bool bounded = PAIR || UNIT || (FLOWGRAD && sp.grad_bound != nullptr);
    if (bounded && !PAIR && !UNIT) mx = __float_as_uint(sp.grad_bound[b]) & 0x7fffffffu;
    float ucoef = 0.0f;
    if constexpr (UNIT) {
        // the image's coefficient (pair_consist_backward_tiles_kernel's): every gradient of the image is unit * ucoef, and
        // |unit| <= unit_max rounds to at most fl(unit_max * |ucoef|) (rounding is monotone): the bound of the fixed point.
        // A zero coefficient leaves the image without a gradient whatever the unit gradient holds.
        const float* gl = (b >= sp.split) ? sp.gl_fwd : sp.gl_bwd;
        ucoef = (gl || sp.gl_sum || sp.gl_mean) ? u_gl / ((u_cnt == 0.0f) ? 1.0f : u_cnt) : 0.0f;
        if (ucoef != 0.0f) mx = __float_as_uint(u_max * ucoef) & 0x7fffffffu;
    }
    if constexpr (PAIR) {
        // ... PAIR: the pair loss's backward for the workgroup's tiles (pair_consist_backward_tiles_kernel's arithmetic, one
        // pixel per thread, two tiles at a time), times the epilogue masks as st_load_grad forms them, into the stash;
        // the maximum comes out on the way.  A final flow with a zero x component (everything the masks or the
        // occlusion check removed) has a zero gradient: no taps.
        const int rem1 = n_hits - part * ST_WAVES;
        const int n_mine = (rem1 / (G * ST_WAVES)) * ST_WAVES + min(rem1 % (G * ST_WAVES), ST_WAVES);
        const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
        const int dir = b >= sp.split ? 1 : 0, pb = b - dir * sp.split;
        const float cnt = sp.sums[pb * 4 + (dir ? 1 : 3)];
        const float* gl = dir ? sp.gl_fwd : sp.gl_bwd;
        const float coef = gl ? gl[pb] / ((cnt == 0.0f) ? 1.0f : cnt) : 0.0f;
        const float* src = dir ? sp.image_ref : sp.image;
        const float* tgt = dir ? sp.image : sp.image_ref;
        const float* jit = dir ? sp.jitter : sp.jitter_ref;
        const int64_t hw_img = (int64_t)sp.H * sp.W;
        for (int m = half; m < n_mine; m += 2) {
            const int t = tile_at(part * ST_WAVES + (m % ST_WAVES) + (m / ST_WAVES) * G * ST_WAVES);
            const int x = (t % sp.tiles_x) * ST_TW + (tid & 31), ry = (t / sp.tiles_x) * ST_TH + (tid >> 5);
            const int y = is - 1 - ry;
            if (x >= sp.W || ry >= is || y >= sp.H) continue;
            const int64_t pixc = (int64_t)y * sp.W + x;
            float2 gq = make_float2(0.0f, 0.0f);
            // the coverage word, the flow and the epilogue masks of the pixel are requested TOGETHER (all inside their
            // allocations; what an uncovered row pair holds there is never used): the taps are the only dependent trip
            const uint32_t word = sp.tile_hit[(int64_t)b * T + t];
            const int64_t o = ((int64_t)b * is + y) * is + x;
            float2 uv = *reinterpret_cast<const float2*>(sp.flow + ((int64_t)b * hw_img + pixc) * 2);
            float a = sp.m_pre[o];
            float mxo = (b < sp.split ? sp.m_x_lo[o] : sp.m_x_hi[o - (int64_t)sp.split * is * is]), oc = sp.occl[o];
            pin(uv.x); pin(uv.y); pin(a); pin(mxo); pin(oc);
            if (coef != 0.0f && ((word >> (8 * ((ry & 7) >> 1))) & 0xffu) != 0u) {
                if (uv.x != 0.0f) {
                    const float post = mxo * oc;
                    const DirTaps tp = pair_taps(uv, x, y, sp.H, sp.W);
                    DirRaw2 q{};
                    pair_load(tp, src, tgt, jit, jit, sp.Cj, false, pb, pixc, hw_img, q);
                    pin(q);
                    const DirRaw r = unpack(q, tp.a);
                    const DirOut e = pair_eval(tp, r, sp.H, sp.W, sp.pair_thresh, false);
                    const float2 gp = pair_grad(tp, r, e, sp.H, sp.W, coef);
                    gq = make_float2((gp.x * post) * a, (gp.y * post) * a);
                }
            }
            *reinterpret_cast<float2*>(sp.stash + ((int64_t)b * hw_img + pixc) * 2) = gq;
            mx = max(mx, max(__float_as_uint(gq.x) & 0x7fffffffu, __float_as_uint(gq.y) & 0x7fffffffu));
        }
        __threadfence_block();  // the stash is read back by other waves of this workgroup after the barrier below
    }
    for (int h = part * ST_WAVES + wave; h < n_hits && !bounded; h += G * ST_WAVES) {
        const int t = tile_at(h);
        const int yi = (t / sp.tiles_x) * ST_TH + r, x = (t % sp.tiles_x) * ST_TW + x4;
        if (yi >= is || x >= is) continue;  // partial tile at the image border (rows are multiples of 4 wide)
        float g[4][3];
        st_load_grad<FLOWGRAD>(sp, b, yi, x, g);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) mx = max(mx, __float_as_uint(g[j][ch]) & 0x7fffffffu);
    }
    unsigned bm = 0u;
    // (WORK: the barrier behind the table zeroing waits in front of the first LDS atomic of pass 2, behind the tile's loads)
    if constexpr (UNIT) {
        if (!WORK) __syncthreads();  // table zeroed
        MR_ST_STAMP(2);
        bm = mx;          // (the image's bound: the same in every lane of the workgroup, nothing to reduce)
    } else {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
        if (lane == 0) wmax[wave] = mx;
        __syncthreads();  // table zeroed, maxima visible
        MR_ST_STAMP(2);
#pragma unroll
        for (int k = 0; k < ST_WAVES; k++) bm = max(bm, wmax[k]);
    }
    if (bm == 0u) return;                  // every gradient is +-0: nothing to add (block-uniform)
    const bool finite = bm < 0x7f800000u;  // else: fp32 global atomics, Inf / NaN propagate
    // A table cell receives at most 3 terms per pixel (a face may name one vertex three times) of every tile this
    // workgroup walks, each below 2^ST_FIX_BITS in magnitude: 2^13 of them fit the 51-bit field of the sums.  A workgroup
    // with more (beyond ten covered tiles: large rasters, screen-filling meshes) gives up one bit of the 37 per doubling
    // -- at 256 tiles per workgroup that is still 2^-32 of the largest gradient per term.
    const int rem = n_hits - part * ST_WAVES;
    const long long terms = 3LL * ST_TW * ST_TH * ((rem / (G * ST_WAVES)) * ST_WAVES + min(rem % (G * ST_WAVES), ST_WAVES));
    int headroom = 0;
    while ((terms >> headroom) >= (1LL << (ST_SUM_BITS - ST_FIX_BITS))) headroom++;
    const int shift = ST_FIX_BITS - headroom - ((int)(bm >> 23) - 126);
    // Round 5: float -> fixed point by the "magic number" addition, two double-precision instructions per value instead of
    // eight (cvt, ldexp, trunc, ldexp, floor, fma, two conversions: the main pass was bound by them -- 24 values per lane, 8
    // waves per SIMD): v 2^shift + 1.5 x 2^52, rounded to an integer by that very addition, has v's fixed-point value in two's
    // complement in the low 51 bits of its mantissa field and a CONSTANT above them; the raw bit patterns are summed by the
    // 64-bit LDS atomics, the constants pile up above bit 50 and are dropped at the flush (sign extension from bit 50).
    // |terms| < 2^ST_FIX_BITS and at most 2^(ST_SUM_BITS - ST_FIX_BITS) of them per cell: the sum stays inside the field.
    const double fix_scale = ldexp(1.0, shift);
    float* out = p.grad_vcolors + (int64_t)b * p.V * 3;

    // pass 2
    // (WORK: every wave takes a FIRST trip, with or without a tile -- the workgroup's barrier sits inside it, at ONE place for
    // all waves; rounds 5's form had the waves without a tile meet the others at a second barrier behind the loop)
    for (int h = part * ST_WAVES + wave, trip = 0; h < n_hits || (WORK && trip == 0); h += G * ST_WAVES, trip++) {
        const bool live = h < n_hits;
        const int t = live ? tile_at(h) : 0;
        const int yi = (t / sp.tiles_x) * ST_TH + r, x = (t % sp.tiles_x) * ST_TW + x4;
        const bool inside = live && yi < is && x < is;
        int4 f4 = make_int4(-1, -1, -1, -1);
        if (inside) f4 = *reinterpret_cast<const int4*>(fim_b + (int64_t)yi * is + x);
        const int fn[4] = {f4.x, f4.y, f4.z, f4.w};
        // (a tile the coverage bytes name holds a covered pixel, and the wave IS the tile: testing here would only put
        // the face-index round trip in front of all the other loads; without coverage bytes every tile is walked)
        if (!sp.tile_hit && __ballot(fn[0] >= 0 || fn[1] >= 0 || fn[2] >= 0 || fn[3] >= 0) == 0ull) continue;
        float g[4][3], w[4][3], zp[4], vz[4][3];
        int vid[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            zp[j] = 1.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) { g[j][k] = 0.0f; w[j][k] = 0.0f; }
        }
        if (inside) {
            st_load_grad<FLOWGRAD, PAIR, UNIT>(sp, b, yi, x, g, ucoef);
            const float4* wq = reinterpret_cast<const float4*>(sp.weight + (((int64_t)b * is + yi) * is + x) * 3);
            const float4 w0 = wq[0], w1 = wq[1], w2 = wq[2];
            w[0][0] = w0.x; w[0][1] = w0.y; w[0][2] = w0.z; w[1][0] = w0.w; w[1][1] = w1.x; w[1][2] = w1.y;
            w[2][0] = w1.z; w[2][1] = w1.w; w[2][2] = w2.x; w[3][0] = w2.y; w[3][1] = w2.z; w[3][2] = w2.w;
            if (REC) {
                // per-pixel records of the forward: the winner's vertex ids next to its sampling weights -- every load
                // of the pixel group is in flight at once, no index -> vertex chain
                const int4* vq = reinterpret_cast<const int4*>(sp.vid_map + (((int64_t)b * is + yi) * is + x) * 3);
                const int4 v0 = vq[0], v1 = vq[1], v2 = vq[2];
                vid[0][0] = v0.x; vid[0][1] = v0.y; vid[0][2] = v0.z; vid[1][0] = v0.w; vid[1][1] = v1.x; vid[1][2] = v1.y;
                vid[2][0] = v1.z; vid[2][1] = v1.w; vid[2][2] = v2.x; vid[3][0] = v2.y; vid[3][1] = v2.z; vid[3][2] = v2.w;
                // Round 5: the three (vertex, weight) pairs of a pixel are ROTATED by the lane's row in the tile (mod 3).  A
                // face spans three or four rows, so the lanes above one another hold the same three vertex ids, and the LDS
                // atomics of step k below all hit the same table cell -- 65 % of this kernel's LDS-active cycles were such
                // collisions (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r05_pmc_summary.txt), the LDS pipe the
                // longest phase of a workgroup.  Rotated, neighbouring rows add to DIFFERENT vertices of the face in the same
                // step (the sum does not care about the order).  SQ_LDS_BANK_CONFLICT 1.59 -> 1.27 M per launch.  dbg & 4: off.
                if (!(p.dbg & 4)) {
                    // (by row AND column: horizontally adjacent lanes share a face where it straddles their four-pixel groups.
                    // 640 x 640, B = 32, warm: 43.9 us unrotated, 36.7 by row, 33.1 by row + column; at 256 x 256, one tile per
                    // wave, the launch is not bound by its LDS pipe and does not change.  dbg & 8: by row only)
                    const int rot = (p.dbg & 8) ? r % 3 : (r + (lane & 7)) % 3;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int a0 = vid[j][0], a1 = vid[j][1], a2 = vid[j][2];
                        const float b0 = w[j][0], b1 = w[j][1], b2 = w[j][2];
                        vid[j][0] = rot == 0 ? a0 : (rot == 1 ? a1 : a2);
                        vid[j][1] = rot == 0 ? a1 : (rot == 1 ? a2 : a0);
                        vid[j][2] = rot == 0 ? a2 : (rot == 1 ? a0 : a1);
                        w[j][0] = rot == 0 ? b0 : (rot == 1 ? b1 : b2);
                        w[j][1] = rot == 0 ? b1 : (rot == 1 ? b2 : b0);
                        w[j][2] = rot == 0 ? b2 : (rot == 1 ? b0 : b1);
                    }
                }
            } else {
                const float4 d4 = *reinterpret_cast<const float4*>(sp.depth + ((int64_t)b * is + (is - 1 - yi)) * is + x);
                zp[0] = d4.x; zp[1] = d4.y; zp[2] = d4.z; zp[3] = d4.w;
            }
        }
        if (!REC) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool won = fn[j] >= 0;
                const bool o = fn[j] >= p.F0;  // reversed copy of face fn - F0
                const int32_t* ix = fidx_b + (int64_t)(won ? (o ? fn[j] - p.F0 : fn[j]) : 0) * 3;
#pragma unroll
                for (int k = 0; k < 3; k++) vid[j][k] = won ? ix[o ? 2 - k : k] : 0;
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 3; k++) vz[j][k] = fn[j] >= 0 ? verts_b[(int64_t)vid[j][k] * 3 + 2] : 1.0f;
        }
        // (Measured and dropped in round 5: forming the fixed-point value from the float's bits with 64-bit integer shifts
        // instead of the double-precision ldexp + conversion -- main pass 5.2 -> 7.6 us per workgroup, the 64-bit shifts and
        // selects are slower than the conversion sequence; summing a lane's consecutive pixels of one face in registers
        // before the LDS atomic -- exact, a third fewer atomics, but 54 -> 82 vector registers: three workgroups per compute
        // unit instead of four, or 92 bytes of scratch when capped.)
        if (WORK && trip == 0) __syncthreads();  // (table zeroed; the one barrier of pass 2, reached by every wave)
        if (!live) break;
        // (the finite / non-finite choice -- block-uniform -- is made ONCE around the 24 additions, not inside each of them: the
        // fixed-point path is straight-line code the compiler can interleave, 48 basic blocks otherwise)
        auto add_all = [&](auto fin) __attribute__((always_inline)) {
        constexpr bool FIN = decltype(fin)::value;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (fn[j] < 0) continue;
            float wgs[3];
            if (REC) {
                wgs[0] = w[j][0]; wgs[1] = w[j][1]; wgs[2] = w[j][2];
            } else {
                float fv[9];
                fv[2] = vz[j][0]; fv[5] = vz[j][1]; fv[8] = vz[j][2];
                float tif[3];
                tex_coords(w[j], zp[j], fv, 2, p.eps, tif);
                // taps pn = 1, 2, 4 are the texels holding the colours of vertices 0, 1, 2
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int pn = 1 << k;
                    float wg = 1.0f;
#pragma unroll
                    for (int q = 0; q < 3; q++) wg *= ((pn >> q) & 1) ? (tif[q] - 0.0f) : (1.0f - (tif[q] - 0.0f));
                    wgs[k] = wg;
                }
            }
            float val[9];
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) val[k * 3 + ch] = wgs[k] * g[j][ch];
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    if (FLOWGRAD && ch == 2) continue;  // the third plane's gradient is identically zero
                    const float v = val[k * 3 + ch];
                    // (records name the vertex behind tap k themselves; else: the texel layout table)
                    const int vtx = REC ? vid[j][k] : sel3(vid[j][0], vid[j][1], vid[j][2], texel_vertex(p.texel, k, fn[j] >= p.F0));
                    const int cell = vtx * NCH + ch;
                    if constexpr (FIN) {
                        const double d = __builtin_fma((double)v, fix_scale, 6755399441055744.0);
                        if (v != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(&vtab[cell]), (unsigned long long)__double_as_longlong(d));
                    } else if (v != 0.0f) {
                        atomicAdd(&out[vtx * 3 + ch], v);
                    }
                }
        }
        };
        if (finite) add_all(std::true_type{}); else add_all(std::false_type{});
    }
    if (!finite) return;  // block-uniform
    __syncthreads();
    MR_ST_STAMP(3);
    for (int k = threadIdx.x; k < p.V * NCH; k += blockDim.x) {
        const long long tsum = (long long)((unsigned long long)vtab[k] << (64 - ST_SUM_BITS - 1)) >> (64 - ST_SUM_BITS - 1);
        if (tsum != 0) {
            const float v = (float)ldexp((double)tsum, -shift);
            if (v != 0.0f) atomicAdd(&out[(k / NCH) * 3 + k % NCH], v);
        }
    }
    MR_ST_STAMP(4);
}

template <bool FLOWGRAD, bool REC>
__global__ void __launch_bounds__(ST_WAVES * MR_WAVE) scatter_tiles_kernel(ScatterTilesParams sp) {
    scatter_tiles_body<FLOWGRAD, REC, false>(sp);
}

// ... with the pair loss's backward folded into pass 1 (mr_flow_pair_backward_tiles).  Eight waves per SIMD: four of
// these 8-wave workgroups per compute unit, i.e. all 1024 of a 128-image launch resident at once, as for the kernel above.
__global__ void __launch_bounds__(ST_WAVES * MR_WAVE) __attribute__((amdgpu_waves_per_eu(8, 8)))
pair_scatter_tiles_kernel(ScatterTilesParams sp) {
    scatter_tiles_body<true, true, true>(sp);
}

// ... and with the pair loss's gradient already formed by the forward launch (mr_flow_pair_backward_unit_tiles): the
// scatter alone, its gradient = unit gradient x the image's coefficient, its bound = the forward's maximum x |coefficient|
__global__ void __launch_bounds__(ST_WAVES * MR_WAVE) unit_scatter_listing_kernel(ScatterTilesParams sp) {
    scatter_tiles_body<true, true, false, true, false>(sp);
}
// ... and with the workgroups handed out over the forward's covered-tile lists (WORK; what the step launches since round 5)
__global__ void __launch_bounds__(ST_WAVES * MR_WAVE) unit_scatter_tiles_kernel(ScatterTilesParams sp) {
    scatter_tiles_body<true, true, false, true, true>(sp);
}

// ---------------------------------------------------------------------------------------
// E, generic texture size: per-pixel atomics on recomputed sampling weights
// ---------------------------------------------------------------------------------------
template <bool IMG>
__global__ void __launch_bounds__(256) textures_atomic_recompute_kernel(
    const float* __restrict__ faces, const int32_t* __restrict__ fim, const float* __restrict__ grad_rgb,
    float* __restrict__ grad_textures, int64_t npx, int F, int is, int ts, float eps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const int fn = fim[i];
    if (fn < 0) return;
    const int b = (int)(i / ((int64_t)is * is));
    const int pn_ = (int)(i % ((int64_t)is * is));
    const int yi = pn_ / is, xi = pn_ % is;
    Face f;
    load_face(faces + ((int64_t)b * F + fn) * 9, f, is);
    float w[3], zp, tif[3];
    bary(f, xi, yi, zp, w);
    tex_coords(w, zp, f.v, ts, eps, tif);
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; c++) g[c] = grad_rgb[idx3<IMG>(b, yi, xi, c, is)];
    float* gt = grad_textures + ((int64_t)b * F + fn) * ts * ts * ts * 3;
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        float wg; int isc;
        tex_tap(tif, pn, ts, wg, isc);
#pragma unroll
        for (int c = 0; c < 3; c++) atomicAdd(&gt[isc * 3 + c], wg * g[c]);
    }
}

// upstream backward_textures on stored sampling maps (raster NHWC)
__global__ void __launch_bounds__(256) textures_atomic_stored_kernel(
    const int32_t* __restrict__ fim, const float* __restrict__ swgt, const int32_t* __restrict__ sidx,
    const float* __restrict__ grad_rgb, float* __restrict__ grad_textures, int64_t npx, int F, int is,
    int ts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const int fn = fim[i];
    if (fn < 0) return;
    const int64_t b = i / ((int64_t)is * is);
    float* gt = grad_textures + (b * F + fn) * ts * ts * ts * 3;
    const float g[3] = {grad_rgb[i * 3], grad_rgb[i * 3 + 1], grad_rgb[i * 3 + 2]};
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        const float w = swgt[i * 8 + pn];
        const int isc = sidx[i * 8 + pn];
#pragma unroll
        for (int c = 0; c < 3; c++) atomicAdd(&gt[isc * 3 + c], w * g[c]);
    }
}

// upstream backward_depth_map on stored maps (raster orientation)
__global__ void __launch_bounds__(256) depth_atomic_stored_kernel(
    const float* __restrict__ faces, const float* __restrict__ depth_map, const int32_t* __restrict__ fim,
    const float* __restrict__ face_inv_map, const float* __restrict__ weight_map,
    const float* __restrict__ grad_depth, float* __restrict__ grad_faces, int64_t npx, int F, int is) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const int fn = fim[i];
    if (fn < 0) return;
    const int64_t b = i / ((int64_t)is * is);
    const float* face = faces + (b * F + fn) * 9;
    float* gface = grad_faces + (b * F + fn) * 9;
    const float depth = depth_map[i];
    const float d2 = depth * depth;
    const float gd = grad_depth[i];
    const float* inv = face_inv_map + i * 9;
    const float w[3] = {weight_map[i * 3], weight_map[i * 3 + 1], weight_map[i * 3 + 2]};
    const float z[3] = {face[2], face[5], face[8]};
#pragma unroll
    for (int k = 0; k < 3; k++) atomicAdd(&gface[3 * k + 2], gd * w[k] * d2 / (z[k] * z[k]));
    float tmp[2] = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int l = 0; l < 3; l++) tmp[k] += -inv[3 * l + k] / z[l];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int l = 0; l < 2; l++) atomicAdd(&gface[3 * k + l], -gd * tmp[l] * w[k] * d2 * (float)is / 2.0f);
}

// ---------------------------------------------------------------------------------------
// D: NMR pixel-map pseudo-gradient.
// ---------------------------------------------------------------------------------------
struct PixelMapParams {
    const float* faces;
    const int32_t* fim;  // raster orientation
    const float* rgb;
    const float* alpha;
    const float* grad_rgb;
    const float* grad_alpha;
    float* grad_faces;
    int B, F, is;
    float eps;
    int return_rgb, return_alpha;
    int write_backfacing;  // fused path: also zero the rows of culled faces
    int dbg;               // profiling experiments (flags >> 8)
    float* zero_textures;  // nullable: [B*F, 24] texture-gradient rows, zeroed for the faces that own no pixel
    int zero_owner_rows;   // compact_owners_kernel: zero the grad_faces rows of the owning faces too (kernel D by strips adds into them)
};

template <bool IMG>
__device__ __forceinline__ float diff_grad_at(const PixelMapParams& p, int b, int yi, int xi, float a_ref,
                                              const float* rgb_ref) {
    float d = 0.0f;
    const int is = p.is;
    if (p.return_alpha) {
        const int64_t ia = idx1<IMG>(b, yi, xi, is);
        d += (p.alpha[ia] - a_ref) * p.grad_alpha[ia];
    }
    if (p.return_rgb) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int64_t ic = idx3<IMG>(b, yi, xi, k, is);
            d += (p.rgb[ic] - rgb_ref[k]) * p.grad_rgb[ic];
        }
    }
    return d;
}

// One WAVE per face.  The edge / axis / column loops of upstream's per-face walk are wave-uniform
// (every lane evaluates the same scalars); the pixel sweeps along d1 -- up to the image border for
// the "out" sweep -- are spread over the 64 lanes, each lane accumulating partial sums for the six
// (vertex, component) slots, combined by one butterfly reduction at the end.
template <bool IMG>
__global__ void __launch_bounds__(256) pixel_map_kernel(PixelMapParams p) {
    const int64_t total = (int64_t)p.B * p.F;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // face of this wave
    const int lane = threadIdx.x & 63;
    if (i >= total) return;  // wave-uniform
    const int is = p.is;
    const float fis = (float)is;
    const int b = (int)(i / p.F);
    const int fn = (int)(i % p.F);
    float face[9];
#pragma unroll
    for (int k = 0; k < 9; k++) face[k] = p.faces[i * 9 + k];
    const bool front = !backfacing(face);
    if (!front) {
        if (p.write_backfacing && lane < 9) p.grad_faces[i * 9 + lane] = 0.0f;
        return;
    }
    const int32_t* fim_b = p.fim + (int64_t)b * is * is;
    float acc[3][2];  // [vertex][component] partial sums of this lane
#pragma unroll
    for (int k = 0; k < 3; k++) { acc[k][0] = 0.0f; acc[k][1] = 0.0f; }

#pragma unroll 1
    for (int edge_num = 0; edge_num < 3; edge_num++) {
        int pi[3];
        float pp[3][2];
        for (int num = 0; num < 3; num++) pi[num] = (edge_num + num) % 3;
        for (int num = 0; num < 3; num++)
            for (int dim = 0; dim < 2; dim++) pp[num][dim] = 0.5f * (face[3 * pi[num] + dim] * fis + fis - 1.0f);
#pragma unroll 1
        for (int axis = 0; axis < 2; axis++) {
            float q[3][2];
            for (int num = 0; num < 3; num++)
                for (int dim = 0; dim < 2; dim++) q[num][dim] = pp[num][(dim + axis) % 2];
            int direction;
            if (axis == 0)
                direction = (q[0][0] < q[1][0]) ? -1 : 1;
            else
                direction = (q[0][0] < q[1][0]) ? 1 : -1;
            const int d0_from = (int)fmaxf(ceilf(fminf(q[0][0], q[1][0])), 0.0f);
            const int d0_to = (int)fminf(fmaxf(q[0][0], q[1][0]), fis - 1.0f);
            float g0 = 0.0f, g1 = 0.0f;  // this lane's share for vertices pi[0], pi[1], component 1 - axis
            for (int d0 = d0_from; d0 <= d0_to; d0++) {
                const float fd0 = (float)d0;
                const float d1_cross = (q[1][1] - q[0][1]) / (q[1][0] - q[0][0]) * (fd0 - q[0][0]) + q[0][1];
                const int d1_in = (0 < direction) ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);
                const int d1_out = d1_in + direction;
                if (d1_in < 0 || is <= d1_in) continue;
                if (d1_out < 0 || is <= d1_out) continue;
                // pixel (x, y) of (d0, d1): axis 0 -> (d0, d1), axis 1 -> (d1, d0)
                const int xin = axis == 0 ? d0 : d1_in, yin = axis == 0 ? d1_in : d0;
                const int xout = axis == 0 ? d0 : d1_out, yout = axis == 0 ? d1_out : d0;
                float a_in = 0.0f, a_out = 0.0f, rgb_in[3] = {0, 0, 0}, rgb_out[3] = {0, 0, 0};
                if (p.return_alpha) {
                    a_in = p.alpha[idx1<IMG>(b, yin, xin, is)];
                    a_out = p.alpha[idx1<IMG>(b, yout, xout, is)];
                }
                if (p.return_rgb)
                    for (int k = 0; k < 3; k++) {
                        rgb_in[k] = p.rgb[idx3<IMG>(b, yin, xin, k, is)];
                        rgb_out[k] = p.rgb[idx3<IMG>(b, yout, xout, k, is)];
                    }
                const float c0 = (q[1][0] - q[0][0]) / (q[1][0] - fd0);
                const float c1 = (q[1][0] - q[0][0]) / (fd0 - q[0][0]);
                const bool use0 = q[1][0] != fd0, use1 = q[0][0] != fd0;

                // out sweep: pixels beyond the edge, away from the face, lanes over d1
                if (fim_b[yin * is + xin] == fn) {
                    const int d1_limit = (0 < direction) ? is - 1 : 0;
                    const int d1_from = max(min(d1_out, d1_limit), 0);
                    const int d1_to = min(max(d1_out, d1_limit), is - 1);
                    for (int d1 = d1_from + lane; d1 <= d1_to; d1 += MR_WAVE) {
                        const int xi = axis == 0 ? d0 : d1, yi = axis == 0 ? d1 : d0;
                        const float dg = diff_grad_at<IMG>(p, b, yi, xi, a_in, rgb_in);
                        if (dg <= 0) continue;
                        if (use0) {
                            float dist = c0 * ((float)d1 - d1_cross) * 2.0f / fis;
                            dist = (0 < dist) ? dist + p.eps : dist - p.eps;
                            g0 -= dg / dist;
                        }
                        if (use1) {
                            float dist = c1 * ((float)d1 - d1_cross) * 2.0f / fis;
                            dist = (0 < dist) ? dist + p.eps : dist - p.eps;
                            g1 -= dg / dist;
                        }
                    }
                }
                // in sweep: this face's pixels between the edge and the opposite boundary
                {
                    float d0_cross2;
                    if ((fd0 - q[0][0]) * (fd0 - q[2][0]) < 0)
                        d0_cross2 = (q[2][1] - q[0][1]) / (q[2][0] - q[0][0]) * (fd0 - q[0][0]) + q[0][1];
                    else
                        d0_cross2 = (q[1][1] - q[2][1]) / (q[1][0] - q[2][0]) * (fd0 - q[2][0]) + q[2][1];
                    const int d1_limit = (0 < direction) ? (int)ceilf(d0_cross2) : (int)floorf(d0_cross2);
                    const int d1_from = max(min(d1_in, d1_limit), 0);
                    const int d1_to = min(max(d1_in, d1_limit), is - 1);
                    for (int d1 = d1_from + lane; d1 <= d1_to; d1 += MR_WAVE) {
                        const int xi = axis == 0 ? d0 : d1, yi = axis == 0 ? d1 : d0;
                        if (fim_b[yi * is + xi] != fn) continue;
                        const float dg = diff_grad_at<IMG>(p, b, yi, xi, a_out, rgb_out);
                        if (dg <= 0) continue;
                        if (use0) {
                            float dist = c0 * ((float)d1 - d1_cross) * 2.0f / fis;
                            dist = (0 < dist) ? dist + p.eps : dist - p.eps;
                            g0 -= dg / dist;
                        }
                        if (use1) {
                            float dist = c1 * ((float)d1 - d1_cross) * 2.0f / fis;
                            dist = (0 < dist) ? dist + p.eps : dist - p.eps;
                            g1 -= dg / dist;
                        }
                    }
                }
            }
            // slot (vertex pi[0], comp 1 - axis) += g0 ; slot (vertex pi[1], comp 1 - axis) += g1
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (k == pi[0]) acc[k][1 - axis] += g0;
                if (k == pi[1]) acc[k][1 - axis] += g1;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            acc[k][0] += __shfl_xor(acc[k][0], off);
            acc[k][1] += __shfl_xor(acc[k][1], off);
        }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            p.grad_faces[i * 9 + 3 * k + 0] = acc[k][0];
            p.grad_faces[i * 9 + 3 * k + 1] = acc[k][1];
            p.grad_faces[i * 9 + 3 * k + 2] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------
// D by strips: helpers
// ---------------------------------------------------------------------------------------
constexpr int PM_LONG = 12;  // "in" sweeps longer than this are cut into chunk tasks, shorter ones are walked by the lane that found them

// one accepted term of a walk: -dg / (c * (d1 - d1_cross) * 2 / is +- eps), with the two divisions
// done as multiplications by reciprocals (v_rcp_f32, 1 ulp): the walks evaluate > 10^8 of these per
// launch and an IEEE division is ten instructions
__device__ __forceinline__ float pm_term(float dg, float c, float fd1, float d1_cross, float two_over_is, float eps) {
    float dist = (c * two_over_is) * (fd1 - d1_cross);  // c * two_over_is is loop-invariant
    dist = (0 < dist) ? dist + eps : dist - eps;
    return dg * __builtin_amdgcn_rcpf(dist);
}
// owns[b * F + face] = 1 for every face that won a pixel (four pixels per thread)
// (`zero` / `n_zero`: a word array cleared on the way -- the strip weights of kernel D, which the next kernel adds into)
__global__ void __launch_bounds__(256) mark_owners_kernel(const int32_t* __restrict__ fim, uint8_t* __restrict__ owns,
                                                          int64_t npx4, int64_t px_per_image, int F,
                                                          unsigned* __restrict__ zero, int64_t n_zero) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_zero) zero[i] = 0u;
    if (i >= npx4) return;
    const int4 f = reinterpret_cast<const int4*>(fim)[i];
    uint8_t* o = owns + ((i * 4) / px_per_image) * F;
    if (f.x >= 0) o[f.x] = 1;
    if (f.y >= 0) o[f.y] = 1;
    if (f.z >= 0) o[f.z] = 1;
    if (f.w >= 0) o[f.w] = 1;
}

__global__ void __launch_bounds__(256) mark_owners_scalar_kernel(const int32_t* __restrict__ fim,
                                                                 uint8_t* __restrict__ owns, int64_t npx,
                                                                 int64_t px_per_image, int F,
                                                                 unsigned* __restrict__ zero, int64_t n_zero) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_zero) zero[i] = 0u;
    if (i >= npx) return;
    const int f = fim[i];
    if (f >= 0) owns[(i / px_per_image) * F + f] = 1;
}

// the sum of a value over the wave, valid in lane 63: row shifts and the two row broadcasts of the gfx9 DPP unit
// (six VALU instructions, no LDS crossbar)
__device__ __forceinline__ float wave_sum_last(float v) {
    auto step = [](float x, auto ctrl, auto row_mask) {
        return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false));
    };
    v = step(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
    v = step(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
    v = step(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
    v = step(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8 -> lane 15 of a row = the row
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1, 3
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2, 3
    return v;
}
// the maximum of a non-negative value over the wave, valid in lane 63
__device__ __forceinline__ float wave_max_last(float v) {
    auto step = [](float x, auto ctrl, auto row_mask) {
        return fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false)));
    };
    v = step(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
    v = step(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
    v = step(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
    v = step(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
    return v;
}

// The faces that own a pixel (typically a fifth of them), compacted into a list for the gather / the walk kernel; the
// others get their zero rows here.  One workgroup per image and CO_TPB * CO_PER faces of it, ONE global atomic each (two with
// `img_recs`: the front-facing ones once more image by image, [B][F] records of two float4 -- the vertices in pixel
// coordinates, 0.5 * (v * is + is - 1) as kernel D computes them, and the face number -- with the counts in
// img_count[b] and the range of lines their crossings can fall on in img_count[B + 4 b ..]: kernel D by strips walks
// the owners of ONE image and reads nothing else of a face).
// faces per thread: 1 -- seven workgroups per image of the bench meshes instead of two; most of this kernel's time is the zero
// rows it writes (60 MB per launch at the metric config), which 128 workgroups do not stream at the chip's rate: 17.8 -> 13.8 us
#ifndef MR_CO_PER
#define MR_CO_PER 1
#endif
constexpr int CO_TPB = 1024, CO_PER = MR_CO_PER;
__global__ void __launch_bounds__(CO_TPB) compact_owners_kernel(PixelMapParams p, const uint8_t* __restrict__ owns,
                                                                unsigned* __restrict__ counter, uint32_t* __restrict__ list,
                                                                unsigned* __restrict__ img_count, float4* __restrict__ img_recs,
                                                                int chunks, unsigned* __restrict__ strip_w, int strip_l,
                                                                int strips_axis) {
    __shared__ unsigned s_cnt, s_base, s_icnt, s_ibase, s_range[4];
    __shared__ uint8_t s_own[CO_TPB * CO_PER];
    constexpr int CO_WMAX = 2048;         // strip weights of the image gathered in LDS first (2 * strips_axis <= CO_WMAX)
    __shared__ unsigned s_w[CO_WMAX];
    const bool lds_w = strip_w != nullptr && 2 * strips_axis <= CO_WMAX;
    if (lds_w)
        for (int q = threadIdx.x; q < 2 * strips_axis; q += CO_TPB) s_w[q] = 0u;
    const int b = blockIdx.x / chunks, fc0 = (blockIdx.x % chunks) * (CO_TPB * CO_PER);
    const int nf = min(CO_TPB * CO_PER, p.F - fc0);  // faces of this workgroup
    const int64_t f0 = (int64_t)b * p.F + fc0;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) { s_cnt = 0u; s_icnt = 0u; }
    if (tid < 4) s_range[tid] = 0u;
    __syncthreads();
    unsigned pos[CO_PER], ipos[CO_PER];
    bool own[CO_PER], iown[CO_PER];
    float pxy[CO_PER][6];
#pragma unroll
    for (int k = 0; k < CO_PER; k++) {
        iown[k] = false; ipos[k] = 0u;
        const int q = k * CO_TPB + tid;
        const int64_t i = f0 + q;
        own[k] = q < nf && owns[i] != 0;
        // bit 0: owns a pixel; bit 1: its grad_faces row is to be zeroed here
        const bool zero_row = q < nf && (!own[k] || p.zero_owner_rows) && p.grad_faces &&
                              (p.write_backfacing || !backfacing(p.faces + i * 9));
        s_own[q] = (own[k] ? 1 : 0) | (zero_row ? 2 : 0);
        const unsigned long long m = __ballot(own[k]);
        unsigned wbase = 0u;
        if (lane == 0 && m) wbase = atomicAdd(&s_cnt, (unsigned)__popcll(m));
        pos[k] = (unsigned)__shfl((int)wbase, 0) + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (img_recs) {
            // the record, and the lines (columns, rows) the crossings of the face can fall on: the union over its edges
            // of kernel D's d0 ranges, kept per image as {max d0_to + 1, is - min d0_from} per axis (0 = none)
            const float fis = (float)p.is;
            float f9[9];
#pragma unroll
            for (int c = 0; c < 9; c++) f9[c] = own[k] ? p.faces[i * 9 + c] : 0.0f;
            iown[k] = own[k] && !backfacing(f9);
#pragma unroll
            for (int v = 0; v < 3; v++) {
                pxy[k][2 * v] = 0.5f * (f9[3 * v] * fis + fis - 1.0f);
                pxy[k][2 * v + 1] = 0.5f * (f9[3 * v + 1] * fis + fis - 1.0f);
            }
            float hi[2] = {0.0f, 0.0f}, lo[2] = {0.0f, 0.0f};
#pragma unroll
            for (int axis = 0; axis < 2; axis++) {
                const float q0 = pxy[k][axis], q1 = pxy[k][2 + axis], q2 = pxy[k][4 + axis];
                const int from = (int)fmaxf(ceilf(fminf(fminf(q0, q1), q2)), 0.0f);
                const int to = (int)fminf(fmaxf(fmaxf(q0, q1), q2), fis - 1.0f);
                if (iown[k] && from <= to) {
                    hi[axis] = (float)(to + 1); lo[axis] = (float)(p.is - from);
                    // ... and, strip by strip, how many owners have crossings there: the strips' weights (strip_list_kernel)
                    if (strip_w)
                        for (int sx = from / strip_l; sx <= to / strip_l; sx++) {
                            if (lds_w) atomicAdd(&s_w[axis * strips_axis + sx], 1u);
                            else atomicAdd(&strip_w[((int64_t)b * 2 + axis) * strips_axis + sx], 1u);
                        }
                }
            }
            const unsigned long long mi = __ballot(iown[k]);
            if (mi) {  // (wave-uniform)
                const float r0 = wave_max_last(hi[0]), r1 = wave_max_last(lo[0]), r2 = wave_max_last(hi[1]), r3 = wave_max_last(lo[1]);
                unsigned ibase = 0u;
                if (lane == 63) {
                    ibase = atomicAdd(&s_icnt, (unsigned)__popcll(mi));
                    atomicMax(&s_range[0], (unsigned)r0); atomicMax(&s_range[1], (unsigned)r1);
                    atomicMax(&s_range[2], (unsigned)r2); atomicMax(&s_range[3], (unsigned)r3);
                }
                ipos[k] = (unsigned)__shfl((int)ibase, 63) + (unsigned)__popcll(mi & ((1ull << lane) - 1ull));
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        s_base = atomicAdd(counter, s_cnt);
        if (img_recs) s_ibase = atomicAdd(&img_count[b], s_icnt);
    }
    if (img_recs && tid < 4 && s_range[tid]) atomicMax(&img_count[p.B + 4 * b + tid], s_range[tid]);
    if (lds_w)
        for (int q = tid; q < 2 * strips_axis; q += CO_TPB)
            if (s_w[q]) atomicAdd(&strip_w[(int64_t)b * 2 * strips_axis + q], s_w[q]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CO_PER; k++)
        if (own[k]) {
            list[s_base + pos[k]] = (uint32_t)(f0 + k * CO_TPB + tid);
            if (iown[k]) {
                float4* rec = img_recs + 2 * ((int64_t)b * p.F + s_ibase + ipos[k]);
                rec[0] = make_float4(pxy[k][0], pxy[k][1], pxy[k][2], pxy[k][3]);
                rec[1] = make_float4(pxy[k][4], pxy[k][5], __int_as_float(fc0 + k * CO_TPB + tid), 0.0f);
            }
        }
    // the zero rows of the workgroup's faces (36 bytes of grad_faces, 96 bytes of texture gradient), consecutive
    // lanes on consecutive addresses
    if (p.grad_faces) {
        float* rows = p.grad_faces + f0 * 9;
        for (int q = tid; q < nf * 9; q += CO_TPB)
            if (s_own[q / 9] & 2) rows[q] = 0.0f;
    }
    if (p.zero_textures) {
        float4* rows = reinterpret_cast<float4*>(p.zero_textures + f0 * 24);
        for (int q = tid; q < nf * 6; q += CO_TPB)
            if (!(s_own[q / 6] & 1)) rows[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// ---------------------------------------------------------------------------------------
// D by STRIPS: the sweeps read the maps from LDS, every map byte is fetched once per axis
// ---------------------------------------------------------------------------------------
// The walks of kernel D are sweeps along image columns (axis 0) and rows (axis 1): 1.5 x 10^8 terms per launch of the
// bench workload, every term the eight map values of a pixel.  Walking face by face (pixel_map_kernel; rounds 1-2 also
// on two packed copies of the maps, 2 x 32 B per pixel of workspace and a pack pass in front) re-reads a line once per
// face that crosses it -- dozens of times -- from L2 or beyond.  Here a workgroup OWNS a strip of L adjacent lines of
// one image and one axis: it stages the strip's values in LDS once (for axis 0 straight from the planes, L * 4 bytes
// per image row: the other columns of those cache lines are the neighbouring strips', which run next to it on the same
// XCD -- all strips of an image do, blockIdx -> (image, strip) below -- so they come from that XCD's L2), lists the
// (face, edge) pairs of the image's owning faces whose crossings fall into the strip, and runs their sweeps out of LDS.
//   * owners: compact_owners_kernel leaves, per image, the front-facing owners as records of pixel-space vertices and
//     the range of lines their crossings can fall on; a strip outside the range returns after one load;
//   * enumeration: a thread per owner, 256 a round (the records of the first rounds requested before the strip is
//     staged, all of a workgroup's global loads in flight together), entries (owner | edge | first line | lines - 1)
//     appended to an LDS queue, flushed when another round might not fit;
//   * waves take batches of 64 / L entries, one LANE per (entry, line): crossing, "in" / "out" pixels, short "in"
//     sweep -- the per-item header of upstream's walk, same arithmetic;
//   * the "out" sweeps and the long "in" sweeps are cut into chunks of 16 positions and handed out one per LANE: the
//     lane fetches its item's header from the lane that computed it (ds_bpermute) and walks the chunk by itself, its
//     start rotated so that the 16 lanes of a ds_read_b128 group never share a bank; two positions' values requested
//     while the previous two are worked on; both distances of a term in packed fp32 and one v_rcp_f32 for the two;
//   * sums: a workgroup contributes to ONE component (1 - axis) of a face's three vertices: chunk sums meet in LDS
//     atomics per item, the lines of an entry in a shuffle, one pair of global atomic adds per entry (rows zeroed by
//     compact_owners_kernel).  Float atomics: the order of the additions across chunks and strips is not fixed -- as
//     for the texture / depth terms of this backward pass upstream; MR_FLAG_REFERENCE_ALGO keeps the ordered walk.
// Measured (B = 64, 256 x 256, 3076 faces, MI355X): 310 us + 21 us (owner records) + 6 us (flags), against 425 us
// + 95 us (pack) + 15 us of the packed per-face walk; ~70 % of it is VALU issue of the 1.5 x 10^8 terms (31
// instructions a term), the rest the per-strip phases that wait on memory.  Profiling switches (flags >> 8): 1 no
// chunk tasks, 2 no short "in" sweeps, 4 enumeration only, 8 tasks without their steps.
#ifdef MR_WG_TIMELINE
__device__ unsigned long long mr_dbg_ps[8192 * 16];  // profiling builds: phase stamps and counts of the first 8192 strips
#define MR_PS_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) mr_dbg_ps[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define MR_PS_COUNT(k, n) do { if (lane == 0) atomicAdd(&s_dbg[k], (unsigned)(n)); } while (0)
#else
#define MR_PS_STAMP(k) do { } while (0)
#define MR_PS_COUNT(k, n) do { } while (0)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
// sum over the channels of (value - reference) * gradient: v = (alpha, r, g, b), g = their gradients (channels a launch
// does not render are staged as zeros); two packed subtractions, a packed multiply and a packed fma
__device__ __forceinline__ float ps_diff_grad(const float4 v, const float4 g, const float4 ref) {
#pragma clang fp contract(fast)  // D is compared to 1e-4, not bit for bit
    const f32x2 v0 = {v.x, v.y}, v1 = {v.z, v.w}, r0 = {ref.x, ref.y}, r1 = {ref.z, ref.w};
    const f32x2 g0 = {g.x, g.y}, g1 = {g.z, g.w};
    const f32x2 d = (v0 - r0) * g0 + (v1 - r1) * g1;
    return d.x + d.y;
}
__device__ __forceinline__ float ps_bound_k(float k) { return (k == k) ? fminf(fmaxf(k, -1e15f), 1e15f) : k; }
// Terms of the walk (profiling switch flags >> 8 & 1024; read back by mr_pixel_map_terms): one term = one evaluation of upstream's
// sweep body (rasterize.py:269-281 -> backward_pixel_map: every position of an "out" sweep, every position of an "in" sweep whose
// pixel belongs to the face) -- the unit bench.py's `d_e_f.frac_of_algorithmic_issue` prices at 10 lane-instructions.
__device__ unsigned long long mr_pixel_map_terms_counter;
__device__ __forceinline__ void ps_count_terms(int n) {
    int tot = n;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&mr_pixel_map_terms_counter, (unsigned long long)tot);
}
constexpr int PS_T = 256;
constexpr int PS_QCAP = 1024;  // queued entries; a round of PS_T faces adds at most 3 * PS_T

// lines per strip: the most whose records leave room for three workgroups per compute unit; 0 = raster too wide
static int strip_lines(int is) {
    for (int L = 4; L >= 1; L >>= 1)
        if (L * (is + 1) * 36LL <= 37 * 1024) return L;
    return ((is + 1) * 36LL <= 50 * 1024) ? 1 : 0;  // (one workgroup of 64 KB)
}
constexpr int PS_TASKS = 1024;  // chunk tasks of a wave's batch: 64 items, at most 16 chunks each
static int64_t strip_lds_bytes(int is, int L) {
    return 2LL * L * (is + 1) * 16 + (((int64_t)L * (is + 1) * 4 + 15) & ~15LL) + PS_QCAP * 4LL +
           (PS_T / MR_WAVE) * (PS_TASKS * 2LL + MR_WAVE * 8LL);
}

// The strips that have work, XCD by XCD (image b runs on XCD b % 8: all strips of an image behind one L2), HEAVIEST FIRST.
// One workgroup per XCD reads the weights compact_owners_kernel left (owners with crossings per strip), and lists the
// strips with a non-zero weight in three classes -- at least half / a quarter of the largest weight, the rest.  The
// strip kernel takes workgroup i = entry i / 8 of XCD i % 8: no workgroup for the 57 % of the strips outside their image's
// line range (each used to hold one of the chip's 768 slots for a microsecond or two), and the strips through the middle
// of the meshes (up to 100 us each) start first instead of wherever the image order put them (workgroup timeline,
// scripts/dstrip_timeline.py: last start 265 us into a 315 us launch, 650-700 working strips resident of 768).
constexpr int SL_T = 1024;
__global__ void __launch_bounds__(SL_T) strip_list_kernel(const unsigned* __restrict__ strip_w, unsigned* __restrict__ lists,
                                                          unsigned* __restrict__ counts, int B, int S, int cap) {
    __shared__ unsigned s_max, s_n[3], s_at[3];
    const int x = blockIdx.x, tid = threadIdx.x;
    const int n_img = (B - x + 7) / 8;  // images x, x + 8, ...
    const int total = n_img * S;
    if (tid == 0) { s_max = 0u; s_n[0] = s_n[1] = s_n[2] = 0u; }
    __syncthreads();
    unsigned mx = 0u;
    for (int i = tid; i < total; i += SL_T) mx = max(mx, strip_w[(int64_t)((i / S) * 8 + x) * S + i % S]);
    if (mx) atomicMax(&s_max, mx);
    __syncthreads();
    const unsigned hi = s_max / 2u, mid = s_max / 4u;
    auto cls = [&](unsigned w) { return w > hi ? 0 : (w > mid ? 1 : 2); };
    for (int i = tid; i < total; i += SL_T) {
        const unsigned w = strip_w[(int64_t)((i / S) * 8 + x) * S + i % S];
        if (w) atomicAdd(&s_n[cls(w)], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        s_at[0] = 0u; s_at[1] = s_n[0]; s_at[2] = s_n[0] + s_n[1];
        counts[x] = s_n[0] + s_n[1] + s_n[2];
    }
    __syncthreads();
    unsigned* out = lists + (int64_t)x * cap;
    for (int i = tid; i < total; i += SL_T) {
        const int b = (i / S) * 8 + x;
        const unsigned w = strip_w[(int64_t)b * S + i % S];
        if (w) out[atomicAdd(&s_at[cls(w)], 1u)] = (unsigned)(b * S + i % S);
    }
}

template <bool IMG, int L>
__device__ __forceinline__ void strip_body(const PixelMapParams& p, const unsigned* __restrict__ img_count,
                                           const float4* __restrict__ img_recs, int strips_axis,
                                           const unsigned* __restrict__ strip_lists,
                                           const unsigned* __restrict__ strip_counts, int list_cap, const unsigned vbid) {
    extern __shared__ float4 ps_lds[];
    __shared__ unsigned s_qn, s_next;
#ifdef MR_WG_TIMELINE
    __shared__ unsigned s_dbg[12];
    if (threadIdx.x < 12) s_dbg[threadIdx.x] = 0u;
#endif
    MR_PS_STAMP(0);
    constexpr int ENT = MR_WAVE / L;  // entries per batch of a wave
    const int is = p.is, stride = is + 1;  // (+1: the L lines of a staging store fall into different banks)
    float4* recA = ps_lds;                 // [L][stride]  alpha, r, g, b
    float4* recB = recA + L * stride;      // [L][stride]  d alpha, d r, d g, d b
    int* fimL = reinterpret_cast<int*>(recB + L * stride);  // [L][stride]  face index map
    unsigned* queue = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(fimL) + ((L * stride * 4 + 15) & ~15));
    unsigned* sweeps = queue + PS_QCAP;  // (the task lists and the per-item sums of the waves)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx -> XCD blockIdx % 8: every strip of an image on the same XCD; entry blockIdx / 8 of that XCD's list of
    // working strips (heaviest first), `b * S + axis * strips_axis + strip`
    // (vbid: blockIdx.x, or the strip's virtual index when the launch also carries the gather's workgroups -- same residue mod 8)
    const unsigned S = 2u * (unsigned)strips_axis, xcd = vbid & 7u, j = vbid >> 3;
    if (j >= strip_counts[xcd]) return;
    const unsigned ent = strip_lists[(int64_t)xcd * list_cap + j];
    const int b = (int)(ent / S);
    const unsigned sidx = ent % S;
    const int axis = (int)(sidx / (unsigned)strips_axis), l0 = (int)(sidx % (unsigned)strips_axis) * L;
    const int nl = min(L, is - l0);
    const bool ra = p.return_alpha != 0, rr = p.return_rgb != 0;
    const float fis = (float)is, two_over_is = 2.0f / fis;
    const int32_t* fim_b = p.fim + (int64_t)b * is * is;

    auto stage = [&](int line, int d1) {
        const int x = axis == 0 ? l0 + line : d1, y = axis == 0 ? d1 : l0 + line;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ra) {
            const int64_t ia = idx1<IMG>(b, y, x, is);
            v[0] = p.alpha[ia]; v[1] = p.grad_alpha[ia];
        }
        if (rr) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int64_t ic = idx3<IMG>(b, y, x, k, is);
                v[2 + k] = p.rgb[ic]; v[5 + k] = p.grad_rgb[ic];
            }
        }
        recA[line * stride + d1] = make_float4(v[0], v[2], v[3], v[4]);
        recB[line * stride + d1] = make_float4(v[1], v[5], v[6], v[7]);
        fimL[line * stride + d1] = fim_b[y * is + x];
    };
    auto stage_strip = [&]() {
        if (axis == 0) {
            for (int idx = tid; idx < L * is; idx += PS_T)
                if ((idx % L) < nl) stage(idx % L, idx / L);
        } else {
            for (int line = 0; line < nl; line++)
                for (int d1 = tid; d1 < is; d1 += PS_T) stage(line, d1);
        }
    };
    if (tid == 0) { s_qn = 0u; s_next = 0u; }
    const unsigned n_own = img_count[b];
    const float4* own_b = img_recs + 2 * (int64_t)b * p.F;

    const int comp = 1 - axis;
    uint16_t* tasks = reinterpret_cast<uint16_t*>(sweeps) + wave * PS_TASKS;
    float2* acc = reinterpret_cast<float2*>(reinterpret_cast<uint16_t*>(sweeps) + (PS_T / MR_WAVE) * PS_TASKS) + wave * MR_WAVE;
    // a sweep is cut into chunks of CH positions (a power of two >= is / 16: at most 16 chunks per sweep; short chunks
    // fill the lanes of the task rounds: a batch of 64 items has a few hundred)
    int CH = 16;
    while (CH * 16 < is) CH <<= 1;

    auto process = [&](unsigned qn) {
#pragma unroll 1
        for (;;) {
            unsigned base = 0u;
            if (lane == 0) base = atomicAdd(&s_next, (unsigned)ENT);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (base >= qn) break;
            MR_PS_COUNT(3, 1);
            const unsigned ei = base + (unsigned)(lane / L);
            const int loff = lane % L;
            const bool have = ei < qn;
            const unsigned ent = have ? queue[ei] : 0u;
            const int slot = (int)(ent & 0xffffffu), e = (int)((ent >> 24) & 3u);
            const int lfirst = (int)((ent >> 26) & 3u), lcnt = (int)((ent >> 28) & 3u);
            bool valid = have && loff <= lcnt;
            const int line = min(lfirst + loff, L - 1), d0 = l0 + line;
            const float4 v01 = own_b[2 * slot], v2n = own_b[2 * slot + 1];
            const float px[3] = {v01.x, v01.z, v2n.x}, py[3] = {v01.y, v01.w, v2n.y};
            const int fn = __float_as_int(v2n.z);
            const int64_t i = (int64_t)b * p.F + fn;
            // q[num]: vertices e, e+1, e+2, the coordinate pair swapped for axis 1
            float qx[3], qy[3];
#pragma unroll
            for (int num = 0; num < 3; num++) {
                const float vx = (e == 0) ? px[num] : ((e == 1) ? px[(num + 1) % 3] : px[(num + 2) % 3]);
                const float vy = (e == 0) ? py[num] : ((e == 1) ? py[(num + 1) % 3] : py[(num + 2) % 3]);
                qx[num] = axis == 0 ? vx : vy;
                qy[num] = axis == 0 ? vy : vx;
            }
            const int direction = (axis == 0) ? ((qx[0] < qx[1]) ? -1 : 1) : ((qx[0] < qx[1]) ? 1 : -1);
            const float fd0 = (float)d0;
            const float d1_cross = (qy[1] - qy[0]) / (qx[1] - qx[0]) * (fd0 - qx[0]) + qy[0];
            const int d1_in = (0 < direction) ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);
            const int d1_out = d1_in + direction;
            valid = valid && !(d1_in < 0 || is <= d1_in) && !(d1_out < 0 || is <= d1_out);
            const int d1_in_c = min(max(d1_in, 0), is - 1), d1_out_c = min(max(d1_out, 0), is - 1);
            const float4 iA = recA[line * stride + d1_in_c], oA = recA[line * stride + d1_out_c];
            const bool visible = valid && fimL[line * stride + d1_in_c] == fn;
            const float a_in = iA.x, a_out = oA.x;
            const float rgb_in[3] = {iA.y, iA.z, iA.w}, rgb_out[3] = {oA.y, oA.z, oA.w};
            const float c0 = (qx[1] - qx[0]) / (qx[1] - fd0);
            const float c1 = (qx[1] - qx[0]) / (fd0 - qx[0]);
            const bool use0 = qx[1] != fd0, use1 = qx[0] != fd0;
            float d0_cross2;
            if ((fd0 - qx[0]) * (fd0 - qx[2]) < 0)
                d0_cross2 = (qy[2] - qy[0]) / (qx[2] - qx[0]) * (fd0 - qx[0]) + qy[0];
            else
                d0_cross2 = (qy[1] - qy[2]) / (qx[1] - qx[2]) * (fd0 - qx[2]) + qy[2];
            const float lim_f = (0 < direction) ? ceilf(d0_cross2) : floorf(d0_cross2);
            const int d1_limit = (lim_f == lim_f) ? (int)fminf(fmaxf(lim_f, -2.0f), fis + 1.0f) : (int)0x80000000;
            const int in_from = max(min(d1_in, d1_limit), 0), in_to = min(max(d1_in, d1_limit), is - 1);
            const bool long_in = valid && (in_to - in_from) >= PM_LONG;

            float g0 = 0.0f, g1 = 0.0f;  // short "in" sweep: the lane walks its own item
            if (valid && !long_in && !(p.dbg & 2)) {
                for (int d1 = in_from; d1 <= in_to; d1++) {
                    if (fimL[line * stride + d1] != fn) continue;
                    const float dg = ps_diff_grad(recA[line * stride + d1], recB[line * stride + d1],
                                                  make_float4(a_out, rgb_out[0], rgb_out[1], rgb_out[2]));
                    if (dg <= 0) continue;
                    if (use0) g0 -= pm_term(dg, c0, (float)d1, d1_cross, two_over_is, p.eps);
                    if (use1) g1 -= pm_term(dg, c1, (float)d1, d1_cross, two_over_is, p.eps);
                }
            }
            if (p.dbg & 1024) {  // (profiling: the terms of the short "in" sweeps)
                int nt = 0;
                if (valid && !long_in)
                    for (int d1 = in_from; d1 <= in_to; d1++) nt += fimL[line * stride + d1] == fn ? 1 : 0;
                ps_count_terms(nt);
            }
            acc[lane] = make_float2(0.0f, 0.0f);  // the sums of this lane's item over its chunks (LDS atomics of the task lanes)

            // The "out" sweeps of the visible items and the long "in" sweeps, one LANE per chunk of CH positions.  A lane
            // fetches the header of its chunk's item from the lane that computed it (ds_bpermute) and walks the chunk
            // by itself -- no staging of headers, no sums across lanes -- starting at the position that puts its
            // records on the LDS bank quad of its lane number: base + position = lane (mod 16) for all lanes and steps
            // alike, so the 16 lanes of a ds_read_b128 group never share a bank.
            const float k0 = c0 * two_over_is, k1 = c1 * two_over_is;
            const int lim = (0 < direction) ? is - 1 : 0;
            const int o_from = max(min(d1_out, lim), 0), o_to = min(max(d1_out, lim), is - 1);
            const int hdr = line | (e << 4) | ((0 < direction ? 1 : 0) << 6) | ((use0 ? 1 : 0) << 7) | ((use1 ? 1 : 0) << 8);
            const int r_out = o_from | (o_to << 16), r_in = in_from | (in_to << 16);
#pragma unroll 1
            for (int kind = 0; kind < 2; kind++) {  // 0: "out" sweeps, 1: long "in" sweeps (only this face's pixels)
                const bool want = kind == 0 ? visible : long_in;
                const int span = kind == 0 ? o_to - o_from : in_to - in_from;
                const int n = want ? span / CH + 1 : 0;
                const int incl = (int)wave_sum_last((float)n);  // (inclusive scan; exact: at most 1024)
                const int total = __builtin_amdgcn_readlane(incl, 63);
                if (total == 0 || (p.dbg & 1)) continue;
                for (int c = 0; c < n; c++) tasks[incl - n + c] = (uint16_t)(lane | (c << 6));
                __builtin_amdgcn_wave_barrier();
#pragma unroll 1
                for (int t0 = 0; t0 < total; t0 += MR_WAVE) {
                    const bool mine = t0 + lane < total;
                    const int task = mine ? (int)tasks[t0 + lane] : 0;
                    const int src = task & 63, c = task >> 6;
                    // (the two distances of a term share ONE reciprocal, 1 / (x y): |k| is kept below 1e15 so that the
                    // product stays finite -- an edge that all but touches the line has k up to inf, its terms are 0 in
                    // upstream's dg / dist and < 1e-15 dg here; a NaN stays a NaN)
                    const float t_cross = __shfl(d1_cross, src), t_k0 = ps_bound_k(__shfl(k0, src)), t_k1 = ps_bound_k(__shfl(k1, src));
                    const int t_hdr = __shfl(hdr, src), t_range = __shfl(kind == 0 ? r_out : r_in, src);
                    const float4 ref = make_float4(__shfl(kind == 0 ? a_in : a_out, src), __shfl(kind == 0 ? rgb_in[0] : rgb_out[0], src),
                                                   __shfl(kind == 0 ? rgb_in[1] : rgb_out[1], src),
                                                   __shfl(kind == 0 ? rgb_in[2] : rgb_out[2], src));
                    const int t_line = t_hdr & 15;
                    const bool t_use0 = (t_hdr >> 7) & 1, t_use1 = (t_hdr >> 8) & 1;
                    const int from = (t_range & 0xffff) + c * CH, to = min(t_range >> 16, from + CH - 1);
                    const int cnt = mine ? to - from + 1 : 0;
                    const int base = t_line * stride + from;
                    const int rot = (lane - base) & 15;
                    const int t_fn = __shfl(fn, src);
                    MR_PS_COUNT(1, 1);
                    MR_PS_COUNT(2, cnt);
                    if (p.dbg & 1024) {  // (profiling: "out" sweeps count every position, long "in" sweeps the face's own pixels)
                        int nt = kind == 0 ? cnt : 0;
                        if (kind == 1)
                            for (int q_ = 0; q_ < cnt; q_++) nt += fimL[base + q_] == t_fn ? 1 : 0;
                        ps_count_terms(nt);
                    }
                    f32x2 ww = {0.0f, 0.0f};
                    // an unused term gets the distance 1 and the weight 0
                    const f32x2 kk = {t_use0 ? t_k0 : 0.0f, t_use1 ? t_k1 : 0.0f};
                    // The records of PG steps are requested together, the next PG while these are worked on (the
                    // compiler keeps the steps of an unrolled loop one behind the other, each waiting for its own LDS
                    // round trip).  No branch in a step: positions behind the chunk's end are read -- LDS, inside the
                    // workgroup's allocation -- and weighted 0.
                    constexpr int PG = 2;  // steps requested together
                    struct Quad { float4 a[PG], g[PG]; int owner[PG], pos[PG]; };
                    auto request = [&](Quad& q, int it0) {
#pragma unroll
                        for (int u = 0; u < PG; u++) {
                            const int pos = (rot + it0 + u) & (CH - 1);
                            q.pos[u] = pos;
                            q.a[u] = recA[base + pos];
                            q.g[u] = recB[base + pos];
                            q.owner[u] = kind == 0 ? 0 : fimL[base + pos];
                        }
                    };
                    if (kind == 0) {
                        // d1 - d1_cross keeps its sign beyond the edge: the +-eps of `dist` is a constant of the sweep.
                        // Both distances of a step at once (packed fp32) and ONE reciprocal for the two,
                        // 1 / (x y) * y and 1 / (x y) * x: v_rcp_f32 is a quarter-rate instruction
                        const float fdir = ((t_hdr >> 6) & 1) ? 1.0f : -1.0f;
                        const f32x2 ee = {t_use0 ? ((0.0f < t_k0 * fdir) ? p.eps : -p.eps) : 1.0f,
                                          t_use1 ? ((0.0f < t_k1 * fdir) ? p.eps : -p.eps) : 1.0f};
                        auto work = [&](const Quad& q) {
#pragma unroll
                            for (int u = 0; u < PG; u++) {
                                float dg = ps_diff_grad(q.a[u], q.g[u], ref);
                                dg = (dg <= 0.0f) ? 0.0f : dg;  // (a NaN stays a NaN, as with upstream's `if (dg <= 0) continue`)
                                dg = q.pos[u] < cnt ? dg : 0.0f;
                                // (the distance of a position behind the chunk's end is taken at the end: beyond it the
                                // sweep may cross the edge, where k * tt + e passes through 0 and 0 * inf would be a NaN)
                                const float tt = (float)(from + min(q.pos[u], cnt - 1)) - t_cross;
                                const f32x2 tt2 = {tt, tt};
                                const f32x2 dist = __builtin_elementwise_fma(kk, tt2, ee);
                                const float r = __builtin_amdgcn_rcpf(dist.x * dist.y);
                                const f32x2 inv = {r * dist.y, r * dist.x};
                                const f32x2 mdg = {-dg, -dg};
                                ww = __builtin_elementwise_fma(mdg, inv, ww);
                            }
                        };
                        Quad x, y;
                        request(x, 0);
#pragma unroll 1
                        for (int it0 = 0; it0 < ((p.dbg & 8) ? 0 : CH); it0 += 2 * PG) {
                            request(y, it0 + PG);
                            work(x);
                            request(x, it0 + 2 * PG);  // (behind the last step: positions wrap inside the chunk, the values are not used)
                            work(y);
                        }
                    } else {
                        auto work = [&](const Quad& q) {
#pragma unroll
                            for (int u = 0; u < PG; u++) {
                                float dg = ps_diff_grad(q.a[u], q.g[u], ref);
                                dg = (dg <= 0.0f) ? 0.0f : dg;
                                dg = ((q.pos[u] < cnt) & (q.owner[u] == t_fn)) ? dg : 0.0f;
                                const float tt = (float)(from + q.pos[u]) - t_cross;
                                float d0_ = t_k0 * tt, d1_ = t_k1 * tt;
                                const float s0 = (0 < d0_) ? p.eps : -p.eps, s1 = (0 < d1_) ? p.eps : -p.eps;
                                d0_ += s0; d1_ += s1;
                                d0_ = t_use0 ? d0_ : 1.0f;
                                d1_ = t_use1 ? d1_ : 1.0f;
                                const float r = __builtin_amdgcn_rcpf(d0_ * d1_);
                                ww.x = __builtin_fmaf(-dg, r * d1_, ww.x);
                                ww.y = __builtin_fmaf(-dg, r * d0_, ww.y);
                            }
                        };
                        Quad x, y;
                        request(x, 0);
#pragma unroll 1
                        for (int it0 = 0; it0 < ((p.dbg & 8) ? 0 : CH); it0 += 2 * PG) {
                            request(y, it0 + PG);
                            work(x);
                            request(x, it0 + 2 * PG);  // (behind the last step: positions wrap inside the chunk, the values are not used)
                            work(y);
                        }
                    }
                    // The chunks of an item are CONSECUTIVE tasks (at most 16): their sums are added up over the lanes of the
                    // segment and its first lane adds the total to the item's slot with a plain LDS read-modify-write -- an
                    // item has one segment per round, and the rounds of a wave run one behind the other.  (Round 5: these were
                    // two LDS float atomics per round; ds_add_f32 costs ~12 cycles PER ACTIVE LANE on gfx950 -- 770 cycles for
                    // a full wave whatever the addresses, 30 x a ds_add_u64, profiles/r05_valu_rate_probe.txt -- i.e. as much
                    // as the sixteen terms of the chunks themselves.)
                    if (p.dbg & 16) {
                        if (mine) {
                            if (t_use0 && ww.x != 0.0f) unsafeAtomicAdd(&acc[src].x, ww.x);
                            if (t_use1 && ww.y != 0.0f) unsafeAtomicAdd(&acc[src].y, ww.y);
                        }
                    } else {
                        float sx = (mine && t_use0) ? ww.x : 0.0f, sy = (mine && t_use1) ? ww.y : 0.0f;
                        const int seg = mine ? src : -1 - lane;
#pragma unroll
                        for (int off = 1; off < 16; off <<= 1) {
                            // (every shuffle in full-wave control flow: a lane that is masked off hands out zeros)
                            const float ox = __shfl_down(sx, off), oy = __shfl_down(sy, off);
                            const int os = __shfl_down(seg, off);
                            const bool same = lane + off < MR_WAVE && os == seg;
                            sx += same ? ox : 0.0f;
                            sy += same ? oy : 0.0f;
                        }
                        const int prev = __shfl_up(seg, 1);
                        const bool head = mine && (lane == 0 || prev != seg);
                        if (head && (sx != 0.0f || sy != 0.0f)) {
                            float2 a = acc[src];
                            a.x += sx; a.y += sy;
                            acc[src] = a;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();  // (the next kind / batch overwrites the task list)
            }
            // an entry's lines -> one pair of sums -> the face's row (zeroed by compact_owners_kernel)
            g0 += acc[lane].x;
            g1 += acc[lane].y;
#pragma unroll
            for (int o = 1; o < L; o <<= 1) {
                g0 += __shfl_xor(g0, o);
                g1 += __shfl_xor(g1, o);
            }
            if (have && loff == 0) {
                if (g0 != 0.0f) unsafeAtomicAdd(&p.grad_faces[i * 9 + 3 * e + comp], g0);
                if (g1 != 0.0f) unsafeAtomicAdd(&p.grad_faces[i * 9 + 3 * (e == 2 ? 0 : e + 1) + comp], g1);
            }
        }
    };

    // The image's front-facing owners, one per thread and round; the records of the first PRE rounds are requested
    // before the strip is staged, so that all of a workgroup's global loads are in flight together.
    constexpr int PRE = 3;
    float4 pre[PRE][2];
#pragma unroll
    for (int j = 0; j < PRE; j++) {
        const unsigned k = (unsigned)(j * PS_T + tid);
        const unsigned kc = k < n_own ? k : 0u;
        pre[j][0] = own_b[2 * kc]; pre[j][1] = own_b[2 * kc + 1];
    }
    stage_strip();
    __syncthreads();
    MR_PS_STAMP(1);
    for (unsigned k0 = 0, j = 0; k0 < n_own; k0 += PS_T, j++) {
        const unsigned k = k0 + tid;
        float4 v01, v2n;
        if (j < PRE) {
#pragma unroll
            for (int jj = 0; jj < PRE; jj++)
                if ((unsigned)jj == j) { v01 = pre[jj][0]; v2n = pre[jj][1]; }
        } else {
            const unsigned kc = k < n_own ? k : 0u;
            v01 = own_b[2 * kc]; v2n = own_b[2 * kc + 1];
        }
        if (k < n_own) {
            const float q[3] = {axis == 0 ? v01.x : v01.y, axis == 0 ? v01.z : v01.w, axis == 0 ? v2n.x : v2n.y};
            unsigned ent[3];
            int n = 0;
#pragma unroll
            for (int e = 0; e < 3; e++) {
                const float q00 = q[e], q10 = q[e == 2 ? 0 : e + 1];
                const int d0_from = (int)fmaxf(ceilf(fminf(q00, q10)), 0.0f);
                const int d0_to = (int)fminf(fmaxf(q00, q10), fis - 1.0f);
                const int lo = max(d0_from, l0), hi = min(d0_to, l0 + L - 1);
                const bool hit = lo <= hi;
                ent[e] = hit ? (k | ((unsigned)e << 24) | ((unsigned)(lo - l0) << 26) | ((unsigned)(hi - lo) << 28)) : 0xffffffffu;
                n += hit ? 1 : 0;
            }
            if (n) {
                unsigned at = atomicAdd(&s_qn, (unsigned)n);
#pragma unroll
                for (int e = 0; e < 3; e++)
                    if (ent[e] != 0xffffffffu) queue[at++] = ent[e];
            }
        }
        __syncthreads();
        const unsigned qn = s_qn;
        if (qn + 3u * PS_T > (unsigned)PS_QCAP || k0 + PS_T >= n_own) {
            MR_PS_STAMP(2);
            MR_PS_COUNT(0, tid == 0 ? qn : 0);
            if (qn && !(p.dbg & 4)) process(qn);
            __syncthreads();
            if (tid == 0) { s_qn = 0u; s_next = 0u; }
            __syncthreads();
        }
    }
    MR_PS_STAMP(3);
#ifdef MR_WG_TIMELINE
    if (tid < 12 && blockIdx.x < 8192) mr_dbg_ps[blockIdx.x * 16 + 4 + tid] = s_dbg[tid];
#endif
}

template <bool IMG, int L>
__global__ void __launch_bounds__(PS_T) pixel_map_strip_kernel(PixelMapParams p, const unsigned* __restrict__ img_count,
                                                               const float4* __restrict__ img_recs, int strips_axis,
                                                               const unsigned* __restrict__ strip_lists,
                                                               const unsigned* __restrict__ strip_counts, int list_cap) {
    strip_body<IMG, L>(p, img_count, img_recs, strips_axis, strip_lists, strip_counts, list_cap, blockIdx.x);
}

// Kernel D's strips and the E / F gather in ONE launch (round 5).  The two are independent once the owner lists stand -- D adds
// to grad_faces' x / y with atomics, the gather adds its depth terms to the same rows with atomics and writes grad_textures,
// which D never touches -- and they complement each other: the walk is 200+ us of vector arithmetic at three waves per SIMD,
// the gather a latency chain of short workgroups (owner -> face -> probes -> gradients) that waits on memory half of its
// time.  Groups of eight workgroups (one per XCD: the strips' XCD residue is kept) alternate between the two grids while both
// last; a gather workgroup reserves the strips' LDS and registers, which costs it nothing (144 registers: three workgroups
// per compute unit either way).  A first version ran the gather on a second, higher-priority stream (fork / join events):
// 0.375 -> 0.334 ms in a process with two queues, but 0.47 ms inside bench.py's full run -- with the five or more queues that
// process has, every cross-queue dependency waited for a ~50 us scheduling quantum (profiles/r05_def_two_streams_trace.txt).
template <bool IMG, int L, bool TEX, bool DEPTH>
__global__ void __launch_bounds__(PS_T) strip_gather_kernel(PixelMapParams p, const unsigned* __restrict__ img_count,
                                                            const float4* __restrict__ img_recs, int strips_axis,
                                                            const unsigned* __restrict__ strip_lists,
                                                            const unsigned* __restrict__ strip_counts, int list_cap,
                                                            GatherParams g, unsigned gather_groups, unsigned strip_groups) {
    const unsigned grp = blockIdx.x >> 3, r = blockIdx.x & 7u;
    bool gather;
    unsigned vg;
    if (p.dbg & 256) {         // (profiling: the two grids alternate group by group while both last)
        const unsigned pairs = min(gather_groups, strip_groups);
        if (grp < 2u * pairs) { gather = (grp & 1u) != 0u; vg = grp >> 1; }
        else { gather = gather_groups > strip_groups; vg = pairs + (grp - 2u * pairs); }
    } else if (p.dbg & 512) {  // (profiling: the strips first)
        gather = grp >= strip_groups; vg = gather ? grp - strip_groups : grp;
    } else {                   // the gather's workgroups first: short, and its few long ones (faces with large boxes) start early
        gather = grp < gather_groups; vg = gather ? grp : grp - gather_groups;
    }
    if (gather) gather_body<IMG, TEX, DEPTH>(g, vg * 8u + r, gather_groups * 8u);
    else strip_body<IMG, L>(p, img_count, img_recs, strips_axis, strip_lists, strip_counts, list_cap, vg * 8u + r);
}


template <typename K, typename... A>
static int launch1d(K kernel, int64_t n, hipStream_t s, A... args) {
    if (n <= 0) return MR_OK;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return MR_ERR_BADARG;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), 0, s, args...);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

// backward workspace: per-face "owns a pixel" flags | owner count | per-image counts and line ranges | owner list |
// per-image owner records (the gather needs the flags, the count and the list)
struct OwnerList {
    uint8_t* owns;
    unsigned* counter;
    unsigned* img_count;  // [B] owners per image, then [B][4] their line ranges (behind the counter: one memset clears all)
    uint32_t* list;
    float4* img_recs;     // [B][F][2] the front-facing owners image by image (compact_owners_kernel)
    size_t owns_bytes, clear_bytes;
};
static int64_t round256(int64_t n) { return (n + 255) & ~255LL; }
static int64_t owner_list_bytes(int B, int F) {
    return round256((int64_t)B * F) + 256 + round256(20LL * B) + round256((int64_t)B * F * 4) + round256((int64_t)B * F * 32);
}
static OwnerList owner_list(void* workspace, int B, int F) {
    OwnerList o;
    o.owns = (uint8_t*)workspace;
    o.owns_bytes = (size_t)round256((int64_t)B * F);
    o.counter = (unsigned*)(o.owns + o.owns_bytes);
    o.img_count = (unsigned*)(o.owns + o.owns_bytes + 256);
    o.clear_bytes = o.owns_bytes + 256 + (size_t)round256(20LL * B);
    o.list = (uint32_t*)(o.owns + o.clear_bytes);
    o.img_recs = (float4*)(o.owns + o.clear_bytes + round256((int64_t)B * F * 4));
    return o;
}
// flags -> lists (+ zero rows); the workspace's flags and counts must have been cleared (ol.clear_bytes) and marked
static int launch_compact(const PixelMapParams& q, const OwnerList& ol, bool per_image, hipStream_t s,
                          unsigned* strip_w = nullptr, int strip_l = 1, int strips_axis = 0) {
    const int chunks = (q.F + CO_TPB * CO_PER - 1) / (CO_TPB * CO_PER);
    const int64_t blocks = (int64_t)q.B * chunks;
    if (blocks > 0x7fffffffLL) return MR_ERR_BADARG;
    if (blocks == 0) return MR_OK;
    hipLaunchKernelGGL(compact_owners_kernel, dim3((unsigned)blocks), dim3(CO_TPB), 0, s, q, (const uint8_t*)ol.owns, ol.counter,
                       ol.list, per_image ? ol.img_count : (unsigned*)nullptr, per_image ? ol.img_recs : (float4*)nullptr, chunks,
                       strip_w, strip_l, strips_axis);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

// ... + kernel D's strip bookkeeping behind it: weights [B][2 strips_axis], per-XCD lists [8][cap], their lengths [8]
struct StripLists {
    unsigned* weights;
    unsigned* lists;
    unsigned* counts;
    int strips_axis, cap;
    int64_t n_weights;
};
static int strip_lines(int is);
static int64_t strip_list_bytes(int B, int is) {
    const int L = strip_lines(is);
    if (L == 0) return 0;
    const int64_t sa = (is + L - 1) / L, S = 2 * sa, cap = (int64_t)((B + 7) / 8) * S;
    return round256((int64_t)B * S * 4) + round256(8 * cap * 4) + 256;
}
static StripLists strip_lists(void* workspace, int B, int F, int is) {
    StripLists sl{};
    const int L = strip_lines(is);
    sl.strips_axis = (is + L - 1) / L;
    const int64_t S = 2LL * sl.strips_axis;
    sl.cap = (int)(((B + 7) / 8) * S);
    sl.n_weights = (int64_t)B * S;
    char* base = (char*)workspace + owner_list_bytes(B, F);
    sl.weights = (unsigned*)base;
    sl.lists = (unsigned*)(base + round256(sl.n_weights * 4));
    sl.counts = (unsigned*)(base + round256(sl.n_weights * 4) + round256(8LL * sl.cap * 4));
    return sl;
}
static int64_t pixel_map_workspace_bytes(int B, int F, int is) {
    return owner_list_bytes(B, F) + strip_list_bytes(B, is);
}

// kernel D by strips needs the workspace, a raster whose lines fit LDS and 24-bit owner numbers
static bool strips_apply(int B, int F, int is, const void* workspace, int64_t workspace_bytes, int flags) {
    return workspace && workspace_bytes >= pixel_map_workspace_bytes(B, F, is) && !(flags & MR_FLAG_REFERENCE_ALGO) &&
           strip_lines(is) != 0 && (int64_t)B * F < 0x7fffffffLL && F < (1 << 24);
}

// kernel D: by strips when a workspace of pixel_map_workspace_bytes is available (leaves the owner list for the gather
// behind), else the plane-reading per-face walk (same results up to the order of the fp32 additions)
// `fused` (nullable): the E / F gather of the same call, to go out INSIDE the strip kernel's launch (strip_gather_kernel);
// *fused_done says whether it did (not when the walk takes the plane-reading kernel)
struct FusedGather {
    GatherParams g;
    bool tex, depth;
    int64_t threads;
};
template <bool IMG>
static int launch_pixel_map(const PixelMapParams& p, void* workspace, int64_t workspace_bytes, int flags,
                            hipStream_t s, const FusedGather* fused = nullptr, bool* fused_done = nullptr) {
    const int64_t nfaces = (int64_t)p.B * p.F;
    if (!strips_apply(p.B, p.F, p.is, workspace, workspace_bytes, flags))
        return launch1d(pixel_map_kernel<IMG>, nfaces * MR_WAVE, s, p);
    const OwnerList ol0 = owner_list(workspace, p.B, p.F);
    const int strip_l = strip_lines(p.is);
    hipError_t e0 = hipMemsetAsync(ol0.owns, 0, ol0.clear_bytes, s);  // flags and the counts behind them
    if (e0 != hipSuccess) return (int)e0;
    const int64_t ppi = (int64_t)p.is * p.is, npx = ppi * p.B;
    const StripLists sl = strip_lists(workspace, p.B, p.F, p.is);
    // (the marking pass clears the weights, one word per thread: 2 B is / L words -- more than B is^2 / 4 pixel quads for
    // rasters of one or two pixels, found by tests/fuzz_parity.py: the launch then covers the weights, the kernels guard)
    int rc = (ppi % 4 == 0) ? launch1d(mark_owners_kernel, std::max(npx / 4, sl.n_weights), s, p.fim, ol0.owns, npx / 4, ppi, p.F, sl.weights, sl.n_weights)
                            : launch1d(mark_owners_scalar_kernel, std::max(npx, sl.n_weights), s, p.fim, ol0.owns, npx, ppi, p.F, sl.weights, sl.n_weights);
    if (rc != MR_OK || nfaces == 0) return rc;
    PixelMapParams q = p;
    q.zero_owner_rows = 1;
    rc = launch_compact(q, ol0, true, s, sl.weights, strip_l, sl.strips_axis);
    if (rc != MR_OK) return rc;
    const int strips_axis = sl.strips_axis;
    hipLaunchKernelGGL(strip_list_kernel, dim3(8), dim3(SL_T), 0, s, (const unsigned*)sl.weights, sl.lists, sl.counts, p.B,
                       2 * strips_axis, sl.cap);
    MR_CHECK_LAUNCH();
    const int64_t grid = 8LL * sl.cap;
    if (grid > 0x7fffffffLL) return MR_ERR_BADARG;
    const size_t lds = (size_t)strip_lds_bytes(p.is, strip_l);
    if (fused && (fused->tex || fused->depth)) {
        const int64_t gblocks = (fused->threads + PS_T - 1) / PS_T;
        const int64_t ggroups = (gblocks + 7) / 8, total = (ggroups + sl.cap) * 8;
        if (total <= 0x7fffffffLL && ggroups <= 0x0fffffffLL) {
            const GatherParams g = fused->g;
#define MR_SG_LAUNCH(L_, T_, D_)                                                                                          \
            hipLaunchKernelGGL((strip_gather_kernel<IMG, L_, T_, D_>), dim3((unsigned)total), dim3(PS_T), lds, s, p,      \
                               (const unsigned*)ol0.img_count, (const float4*)ol0.img_recs, strips_axis,                  \
                               (const unsigned*)sl.lists, (const unsigned*)sl.counts, sl.cap, g, (unsigned)ggroups,       \
                               (unsigned)sl.cap)
#define MR_SG_PICK(L_)                                                                                                    \
            do {                                                                                                          \
                if (fused->tex && fused->depth) MR_SG_LAUNCH(L_, true, true);                                             \
                else if (fused->tex) MR_SG_LAUNCH(L_, true, false);                                                       \
                else MR_SG_LAUNCH(L_, false, true);                                                                       \
            } while (0)
            if (strip_l == 4) MR_SG_PICK(4);
            else if (strip_l == 2) MR_SG_PICK(2);
            else MR_SG_PICK(1);
#undef MR_SG_PICK
#undef MR_SG_LAUNCH
            MR_CHECK_LAUNCH();
            if (fused_done) *fused_done = true;
            return MR_OK;
        }
    }
    auto kernel = strip_l == 4 ? pixel_map_strip_kernel<IMG, 4> : (strip_l == 2 ? pixel_map_strip_kernel<IMG, 2> : pixel_map_strip_kernel<IMG, 1>);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(PS_T), lds, s, p, (const unsigned*)ol0.img_count,
                       (const float4*)ol0.img_recs, strips_axis, (const unsigned*)sl.lists, (const unsigned*)sl.counts, sl.cap);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

}  // namespace mr

using namespace mr;

extern "C" int64_t mr_render_backward_list_workspace_bytes(int batch_size, int num_faces) {
    if (batch_size < 0 || num_faces < 0) return MR_ERR_BADARG;
    return owner_list_bytes(batch_size, num_faces);
}

extern "C" int64_t mr_render_backward_workspace_bytes(int batch_size, int num_faces, int image_size) {
    if (batch_size < 0 || num_faces < 0 || image_size <= 0) return MR_ERR_BADARG;
    return pixel_map_workspace_bytes(batch_size, num_faces, image_size);
}

extern "C" int mr_backward_pixel_map(const float* faces, const int32_t* face_index_map,
                                     const float* rgb_map, const float* alpha_map,
                                     const float* grad_rgb_map, const float* grad_alpha_map,
                                     float* grad_faces, int batch_size, int num_faces, int image_size,
                                     float eps, int return_rgb, int return_alpha, mr_stream_t stream) {
    if (!faces || !face_index_map || !grad_faces) return MR_ERR_BADARG;
    if (return_rgb && (!rgb_map || !grad_rgb_map)) return MR_ERR_BADARG;
    if (return_alpha && (!alpha_map || !grad_alpha_map)) return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || image_size <= 0) return MR_ERR_BADARG;
    if (!return_rgb && !return_alpha) return MR_OK;
    PixelMapParams p{faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, grad_faces,
                     batch_size, num_faces, image_size, eps, return_rgb, return_alpha, 0, 0};
    if (batch_size == 0 || num_faces == 0) return MR_OK;
    // scratch for the walk by strips, stream-ordered like the forward entry point's record list;
    // if it cannot be had the plane-reading kernel does the same job
    hipStream_t s = (hipStream_t)stream;
    void* work = nullptr;
    const int64_t bytes = pixel_map_workspace_bytes(batch_size, num_faces, image_size);
    if (hipMallocAsync(&work, (size_t)bytes, s) != hipSuccess) {
        (void)hipGetLastError();
        work = nullptr;
    }
    const int rc = launch_pixel_map<false>(p, work, work ? bytes : 0, 0, s);
    if (work) (void)hipFreeAsync(work, s);
    return rc;
}

extern "C" int mr_backward_textures(const int32_t* face_index_map, const float* sampling_weight_map,
                                    const int32_t* sampling_index_map, const float* grad_rgb_map,
                                    float* grad_textures, int batch_size, int num_faces, int image_size,
                                    int texture_size, mr_stream_t stream) {
    if (!face_index_map || !sampling_weight_map || !sampling_index_map || !grad_rgb_map || !grad_textures)
        return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || image_size <= 0 || texture_size < 2) return MR_ERR_BADARG;
    const int64_t npx = (int64_t)batch_size * image_size * image_size;
    return launch1d(textures_atomic_stored_kernel, npx, (hipStream_t)stream, face_index_map,
                    sampling_weight_map, sampling_index_map, grad_rgb_map, grad_textures, npx, num_faces,
                    image_size, texture_size);
}

extern "C" int mr_backward_depth_map(const float* faces, const float* depth_map,
                                     const int32_t* face_index_map, const float* face_inv_map,
                                     const float* weight_map, const float* grad_depth_map,
                                     float* grad_faces, int batch_size, int num_faces, int image_size,
                                     mr_stream_t stream) {
    if (!faces || !depth_map || !face_index_map || !face_inv_map || !weight_map || !grad_depth_map ||
        !grad_faces)
        return MR_ERR_BADARG;
    if (batch_size < 0 || num_faces < 0 || image_size <= 0) return MR_ERR_BADARG;
    const int64_t npx = (int64_t)batch_size * image_size * image_size;
    return launch1d(depth_atomic_stored_kernel, npx, (hipStream_t)stream, faces, depth_map, face_index_map,
                    face_inv_map, weight_map, grad_depth_map, grad_faces, npx, num_faces, image_size);
}

extern "C" int mr_render_backward(const float* faces, const float* textures,
                                  const int32_t* face_index_map, const float* rgb_img,
                                  const float* alpha_img, const float* grad_rgb_img,
                                  const float* grad_alpha_img, const float* grad_depth_img,
                                  float* grad_faces, float* grad_textures, void* workspace,
                                  int64_t workspace_bytes, int batch_size, int num_faces,
                                  int image_size, int texture_size, float near_, float far_, float eps,
                                  int return_rgb, int return_alpha, int return_depth, int flags,
                                  mr_stream_t stream) {
    (void)textures; (void)near_; (void)far_;
    if (batch_size < 0 || num_faces < 0 || image_size <= 0) return MR_ERR_BADARG;
    if (batch_size == 0 || num_faces == 0) return MR_OK;
    if (!faces || !face_index_map) return MR_ERR_BADARG;
    if (grad_textures && (!return_rgb || !grad_rgb_img || texture_size < 2)) return MR_ERR_BADARG;
    if (batch_size == 0 || num_faces == 0) return MR_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nfaces = (int64_t)batch_size * num_faces;
    const int64_t npx = (int64_t)batch_size * image_size * image_size;
    int rc = MR_OK;

    // D: pixel-map term (writes all 9 slots of every face row)
    const bool want_d = grad_faces && ((return_rgb && grad_rgb_img && rgb_img) ||
                                       (return_alpha && grad_alpha_img && alpha_img));
    const bool want_f = grad_faces && return_depth && grad_depth_img;
    const bool gather_tex = grad_textures && texture_size == 2 && eps >= 1e-6f && !(flags & 1);
    const bool run_gather = (gather_tex || grad_faces) && (gather_tex || want_f || !want_d);
    // with a workspace the faces that own a pixel are listed once (kernel D needs the list anyway) and the gather
    // walks only those: the others -- four fifths of a hand + object mesh -- get their zero rows when the list is built
    const bool strips_d = strips_apply(batch_size, num_faces, image_size, workspace, workspace_bytes, flags);
    const bool use_list = want_d ? strips_d
                                 : (workspace && workspace_bytes >= owner_list_bytes(batch_size, num_faces) &&
                                    !(flags & MR_FLAG_REFERENCE_ALGO) && nfaces <= 0xffffffffLL);
    // the E / F gather of the faces that own a pixel (or of all of them without a list)
    GatherParams g{};
    g.faces = faces; g.fim = face_index_map;
    g.grad_rgb = gather_tex ? grad_rgb_img : nullptr;
    g.grad_depth = want_f ? grad_depth_img : nullptr;
    g.grad_faces = grad_faces;
    g.grad_textures = gather_tex ? grad_textures : nullptr;
    g.B = batch_size; g.F = num_faces; g.is = image_size; g.eps = eps;
    g.accumulate_faces = want_d ? 1 : 0;
    g.dbg_rows = ((flags >> 8) & 32 ? 1 : 0) | ((flags >> 8) & 64 ? 2 : 0);
    if (use_list) {
        const OwnerList ol = owner_list(workspace, batch_size, num_faces);
        g.owners = ol.list; g.n_owners = ol.counter;
    }
    const int64_t gather_threads = nfaces * GGL;
    auto launch_gather = [&](hipStream_t gs) -> int {
        if (gather_tex && want_f) return launch1d(gather_kernel<true, true, true>, gather_threads, gs, g);
        if (gather_tex) return launch1d(gather_kernel<true, true, false>, gather_threads, gs, g);
        if (want_f) return launch1d(gather_kernel<true, false, true>, gather_threads, gs, g);
        return launch1d(gather_kernel<true, false, false>, gather_threads, gs, g);
    };
    bool gathered = false;
    if (want_d) {
        const int rr = return_rgb && grad_rgb_img && rgb_img, ra = return_alpha && grad_alpha_img && alpha_img;
        PixelMapParams p{faces, face_index_map, rgb_img, alpha_img, grad_rgb_img, grad_alpha_img,
                         grad_faces, batch_size, num_faces, image_size, eps, rr, ra, 1, flags >> 8};
        p.zero_textures = (use_list && run_gather && gather_tex) ? grad_textures : nullptr;
        // Round 5: kernel D's walk and the E / F gather go out as ONE launch when both run over the owner lists
        // (strip_gather_kernel; profiling switch flags >> 8 & 128: two launches, one behind the other, as before)
        FusedGather fg{g, gather_tex, want_f, gather_threads};
        const bool fuse = use_list && strips_d && run_gather && (gather_tex || want_f) && !((flags >> 8) & 128);
        rc = launch_pixel_map<true>(p, workspace, workspace_bytes, flags, s, fuse ? &fg : nullptr, &gathered);
        if (rc != MR_OK) return rc;
    } else if (use_list && run_gather) {
        const OwnerList ol = owner_list(workspace, batch_size, num_faces);
        hipError_t e = hipMemsetAsync(ol.owns, 0, ol.clear_bytes, s);  // flags and the counts behind them
        if (e != hipSuccess) return (int)e;
        const int64_t ppi = (int64_t)image_size * image_size;
        if (ppi % 4 == 0) rc = launch1d(mark_owners_kernel, npx / 4, s, face_index_map, ol.owns, npx / 4, ppi, num_faces, (unsigned*)nullptr, (int64_t)0);
        else rc = launch1d(mark_owners_scalar_kernel, npx, s, face_index_map, ol.owns, npx, ppi, num_faces, (unsigned*)nullptr, (int64_t)0);
        if (rc != MR_OK) return rc;
        PixelMapParams q{};
        q.faces = faces; q.grad_faces = grad_faces; q.B = batch_size; q.F = num_faces; q.is = image_size;
        q.write_backfacing = 1;
        q.zero_textures = gather_tex ? grad_textures : nullptr;
        rc = launch_compact(q, ol, false, s);
        if (rc != MR_OK) return rc;
    }
    if (grad_textures && !gather_tex) {
        const size_t bytes = (size_t)nfaces * texture_size * texture_size * texture_size * 3 * sizeof(float);
        hipError_t e = hipMemsetAsync(grad_textures, 0, bytes, s);
        if (e != hipSuccess) return (int)e;
        rc = launch1d(textures_atomic_recompute_kernel<true>, npx, s, faces, face_index_map, grad_rgb_img,
                      grad_textures, npx, num_faces, image_size, texture_size, eps);
        if (rc != MR_OK) return rc;
    }
    if (run_gather && !gathered) rc = launch_gather(s);
    return rc;
}

extern "C" int mr_render_vc_backward(const float* verts, const int32_t* faces_idx, const int32_t* face_index_map,
                                     const float* weight_map, const float* depth_img, const float* grad_rgb_img,
                                     float* grad_vcolors, int batch_size, int num_verts, int num_faces, int fill_back,
                                     int image_size, float eps, int flags, int texel_layout, mr_stream_t stream) {
    if (batch_size < 0 || num_faces < 0 || num_verts < 0 || image_size <= 0 || !texel_layout_ok(texel_layout)) return MR_ERR_BADARG;
    if (!grad_vcolors && (int64_t)batch_size * num_verts > 0) return MR_ERR_BADARG;
    if (batch_size == 0 || num_verts == 0) return MR_OK;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_vcolors, 0, (size_t)batch_size * num_verts * 3 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    if (num_faces == 0) return MR_OK;
    if (!verts || !faces_idx || !face_index_map || !grad_rgb_img || !(eps >= 1e-6f)) return MR_ERR_BADARG;
    GatherVCParams g{verts, faces_idx, face_index_map, grad_rgb_img, grad_vcolors, batch_size, num_verts, num_faces,
                     fill_back, image_size, eps, flags >> 8, texel_layout};
    // pixel-parallel scatter when the per-image colour table fits LDS (dbg bit 32 forces the gather)
    const int64_t table_bytes = (((int64_t)num_verts * 3 + 1) / 2) * 16;
    if (table_bytes <= SV_MAX_TABLE_BYTES && !(g.dbg & 32)) {
        ScatterVCParams sp{g, weight_map, depth_img, (image_size + SV_W - 1) / SV_W, (image_size + SV_H - 1) / SV_H};
        const int64_t blocks = (int64_t)batch_size * sp.rx_n * sp.ry_n;
        if (blocks > 0x7fffffffLL) return MR_ERR_BADARG;
        const bool stored = weight_map && depth_img && !(g.dbg & 64);
        hipLaunchKernelGGL(stored ? scatter_vc_kernel<true> : scatter_vc_kernel<false>, dim3((unsigned)blocks),
                           dim3(SV_WAVES * MR_WAVE), (size_t)table_bytes, s, sp);
        MR_CHECK_LAUNCH();
        return MR_OK;
    }
    if (image_size > 8192) return MR_ERR_BADARG;  // gather fragment encoding: 13 bits per bbox offset
    return launch1d(gather_vc_kernel<0>, (int64_t)batch_size * num_faces * GLPF, s, g);
}

extern "C" int mr_render_flow_backward(const float* verts, const int32_t* faces_idx, const int32_t* face_index_map,
                                       const uint32_t* tile_hit, const float* weight_map, const float* depth_img,
                                       const float* grad_rgb_img, const float* grad_flow, const float* mask_pre,
                                       const float* mask_x_lo, const float* mask_x_hi, int split, const float* occl,
                                       int height, int width, float* grad_vcolors, int batch_size, int num_verts,
                                       int num_faces, int fill_back, int image_size, float eps, int flags,
                                       const int32_t* vertex_id_map, int texel_layout, const float* grad_bound,
                                       mr_stream_t stream) {
    if (batch_size < 0 || num_faces < 0 || num_verts < 0 || image_size <= 0 || !texel_layout_ok(texel_layout)) return MR_ERR_BADARG;
    if (!grad_vcolors && (int64_t)batch_size * num_verts > 0) return MR_ERR_BADARG;
    if (batch_size == 0 || num_verts == 0) return MR_OK;
    const bool flowgrad = grad_rgb_img == nullptr;
    if (flowgrad && (!grad_flow || !mask_pre || !mask_x_lo || !occl || height <= 0 || width <= 0 || height > image_size ||
                     width > image_size || split < 0 || split > batch_size || (split < batch_size && !mask_x_hi)))
        return MR_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (!(flags & MR_FLAG_OUTPUT_ZEROED)) {  // (the workgroups of an image meet in global atomics on a zeroed output)
        hipError_t e = hipMemsetAsync(grad_vcolors, 0, (size_t)batch_size * num_verts * 3 * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    if (num_faces == 0) return MR_OK;
    if (!face_index_map || !weight_map || !(eps >= 1e-6f)) return MR_ERR_BADARG;
    if (!vertex_id_map && (!verts || !faces_idx || !depth_img)) return MR_ERR_BADARG;
    const int64_t table_bytes = (((int64_t)num_verts * (flowgrad ? 2 : 3) + 1) / 2) * 16;
    // the tile walk reads 4-pixel groups with 16-byte loads and keeps the colour table in LDS
    if (image_size % 4 != 0 || table_bytes > SV_MAX_TABLE_BYTES ||
        (int64_t)((image_size + ST_TW - 1) / ST_TW) * ((image_size + ST_TH - 1) / ST_TH) > ST_MAX_TILES)
        return MR_ERR_NOTIMPL;
    ScatterTilesParams sp{};
    sp.g = GatherVCParams{verts, faces_idx, face_index_map, grad_rgb_img, grad_vcolors, batch_size, num_verts, num_faces,
                          fill_back, image_size, eps, flags >> 8, texel_layout};
    sp.weight = weight_map; sp.depth = depth_img; sp.tile_hit = tile_hit; sp.vid_map = vertex_id_map;
    sp.grad_flow = grad_flow; sp.m_pre = mask_pre; sp.m_x_lo = mask_x_lo; sp.m_x_hi = mask_x_hi; sp.occl = occl;
    sp.split = split; sp.H = height; sp.W = width;
    sp.tiles_x = (image_size + ST_TW - 1) / ST_TW; sp.tiles_y = (image_size + ST_TH - 1) / ST_TH;
    sp.grad_bound = flowgrad ? grad_bound : nullptr;
    sp.groups = ST_G;
    switch ((flags >> 12) & 7) { case 1: sp.groups = 2; break; case 2: sp.groups = 4; break; case 3: sp.groups = 16; break; case 4: sp.groups = 32; break; default: break; }
    const int64_t blocks = (int64_t)batch_size * sp.groups;
    if (blocks > 0x7fffffffLL) return MR_ERR_BADARG;
    auto kernel = vertex_id_map ? (flowgrad ? scatter_tiles_kernel<true, true> : scatter_tiles_kernel<false, true>)
                                : (flowgrad ? scatter_tiles_kernel<true, false> : scatter_tiles_kernel<false, false>);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(ST_WAVES * MR_WAVE), (size_t)table_bytes, s, sp);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_flow_pair_backward_tiles(const int32_t* face_index_map, const uint32_t* tile_hit, const float* weight_map,
                                           const int32_t* vertex_id_map, const float* flows, const float* image_ref,
                                           const float* image, const float* jitter_ref, const float* jitter,
                                           int jitter_channels, const float* sums, const float* grad_loss_fwd,
                                           const float* grad_loss_bwd, const float* mask_pre, const float* mask_x_lo,
                                           const float* mask_x_hi, const float* occl, float* grad_flow_scratch, int height,
                                           int width, float* grad_vcolors, int batch_size, int num_verts, int num_faces,
                                           int fill_back, int image_size, float eps, float pair_thresh, int flags,
                                           int texel_layout, mr_stream_t stream) {
    if (batch_size < 0 || (batch_size & 1) || num_faces < 0 || num_verts < 0 || image_size <= 0 || !texel_layout_ok(texel_layout))
        return MR_ERR_BADARG;
    if (!grad_vcolors && (int64_t)batch_size * num_verts > 0) return MR_ERR_BADARG;
    if (batch_size == 0 || num_verts == 0) return MR_OK;
    if (!flows || !image_ref || !image || !jitter_ref || !jitter || !sums || !grad_loss_fwd || !mask_pre || !mask_x_lo ||
        !mask_x_hi || !occl || !grad_flow_scratch || !tile_hit)
        return MR_ERR_BADARG;
    if (jitter_channels != 1 && jitter_channels != 3) return MR_ERR_BADARG;
    if (height <= 0 || width < 2 || height > image_size || width > image_size || (int64_t)height * width > (1LL << 29))
        return MR_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (!(flags & MR_FLAG_OUTPUT_ZEROED)) {
        hipError_t e = hipMemsetAsync(grad_vcolors, 0, (size_t)batch_size * num_verts * 3 * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    if (num_faces == 0) return MR_OK;
    if (!face_index_map || !weight_map || !vertex_id_map || !(eps >= 1e-6f)) return MR_ERR_BADARG;
    const int64_t table_bytes = (((int64_t)num_verts * 2 + 1) / 2) * 16;
    if (image_size % 4 != 0 || table_bytes > SV_MAX_TABLE_BYTES ||
        (int64_t)((image_size + ST_TW - 1) / ST_TW) * ((image_size + ST_TH - 1) / ST_TH) > ST_MAX_TILES)
        return MR_ERR_NOTIMPL;
    ScatterTilesParams sp{};
    sp.g = GatherVCParams{nullptr, nullptr, face_index_map, nullptr, grad_vcolors, batch_size, num_verts, num_faces,
                          fill_back, image_size, eps, flags >> 8, texel_layout};
    sp.weight = weight_map; sp.tile_hit = tile_hit; sp.vid_map = vertex_id_map;
    sp.m_pre = mask_pre; sp.m_x_lo = mask_x_lo; sp.m_x_hi = mask_x_hi; sp.occl = occl;
    sp.split = batch_size / 2; sp.H = height; sp.W = width;
    sp.tiles_x = (image_size + ST_TW - 1) / ST_TW; sp.tiles_y = (image_size + ST_TH - 1) / ST_TH;
    // workgroups per image: pass 1 (the pair loss's backward) walks a workgroup's tiles two at a time, one behind the other
    // -- larger rasters have more covered tiles per image (a sixth of 256 / 900 / 1600), so they get more workgroups:
    // ~8 tiles each (a 480 x 480 pair at B = 8: 73 -> 46 us with 32 instead of 8)
    sp.groups = ST_G;
    while (sp.groups < 32 && sp.tiles_x * sp.tiles_y > 48 * sp.groups) sp.groups *= 2;
    switch ((flags >> 12) & 7) { case 1: sp.groups = 2; break; case 2: sp.groups = 4; break; case 3: sp.groups = 16; break; case 4: sp.groups = 32; break; case 5: sp.groups = 8; break; default: break; }
    sp.stash = grad_flow_scratch; sp.flow = flows; sp.image_ref = image_ref; sp.image = image; sp.jitter_ref = jitter_ref;
    sp.jitter = jitter; sp.Cj = jitter_channels; sp.sums = sums; sp.gl_fwd = grad_loss_fwd; sp.gl_bwd = grad_loss_bwd;
    sp.pair_thresh = pair_thresh;
    const int64_t blocks = (int64_t)batch_size * sp.groups;
    if (blocks > 0x7fffffffLL) return MR_ERR_BADARG;
    hipLaunchKernelGGL(pair_scatter_tiles_kernel, dim3((unsigned)blocks), dim3(ST_WAVES * MR_WAVE), (size_t)table_bytes, s, sp);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

// mr_flow_pair_backward_unit_tiles with the two extra incoming gradients of mr_pair_step_backward (pair_step.hip): of
// loss_bwd + loss_fwd and of the batch mean (ScatterTilesParams::gl_sum / gl_mean)
int mr_flow_pair_backward_unit_tiles_ex(const int32_t* face_index_map, const uint32_t* tile_hit,
                                        const float* weight_map, const int32_t* vertex_id_map, const float* unit_grad,
                                        const float* unit_grad_max, const float* sums, const float* grad_loss_fwd,
                                        const float* grad_loss_bwd, const float* grad_loss_sum, const float* grad_mean,
                                        int mean_of, int height, int width, float* grad_vcolors,
                                        int batch_size, int num_verts, int num_faces, int fill_back, int image_size,
                                        float eps, int flags, int texel_layout, const void* scatter_work,
                                        mr_stream_t stream) {
    if (batch_size < 0 || (batch_size & 1) || num_faces < 0 || num_verts < 0 || image_size <= 0 || !texel_layout_ok(texel_layout))
        return MR_ERR_BADARG;
    if (!grad_vcolors && (int64_t)batch_size * num_verts > 0) return MR_ERR_BADARG;
    if (batch_size == 0 || num_verts == 0) return MR_OK;
    if (!unit_grad || !unit_grad_max || !sums || !(grad_loss_fwd || grad_loss_sum || grad_mean) || !tile_hit) return MR_ERR_BADARG;
    if (height <= 0 || width < 2 || height > image_size || width > image_size || (int64_t)height * width > (1LL << 29))
        return MR_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (!(flags & MR_FLAG_OUTPUT_ZEROED)) {
        hipError_t e = hipMemsetAsync(grad_vcolors, 0, (size_t)batch_size * num_verts * 3 * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    if (num_faces == 0) return MR_OK;
    if (!face_index_map || !weight_map || !vertex_id_map || !(eps >= 1e-6f)) return MR_ERR_BADARG;
    const int64_t table_bytes = (((int64_t)num_verts * 2 + 1) / 2) * 16;
    if (image_size % 4 != 0 || table_bytes > SV_MAX_TABLE_BYTES ||
        (int64_t)((image_size + ST_TW - 1) / ST_TW) * ((image_size + ST_TH - 1) / ST_TH) > ST_MAX_TILES)
        return MR_ERR_NOTIMPL;
    ScatterTilesParams sp{};
    sp.g = GatherVCParams{nullptr, nullptr, face_index_map, nullptr, grad_vcolors, batch_size, num_verts, num_faces,
                          fill_back, image_size, eps, flags >> 8, texel_layout};
    sp.weight = weight_map; sp.tile_hit = tile_hit; sp.vid_map = vertex_id_map;
    sp.split = batch_size / 2; sp.H = height; sp.W = width;
    sp.tiles_x = (image_size + ST_TW - 1) / ST_TW; sp.tiles_y = (image_size + ST_TH - 1) / ST_TH;
    // workgroups per image: a wave takes a covered tile per round, and larger rasters have more of them per image (a sixth of
    // 256 / 900 / 1600 tiles): 8 / 32 / 32 -- a 480 x 480 pair at B = 8: 30 -> 19 us, 640 x 640 at B = 32: 86 -> 74 us cold
    sp.groups = ST_G;
    while (sp.groups < 32 && sp.tiles_x * sp.tiles_y > 48 * sp.groups) sp.groups *= 2;
    switch ((flags >> 12) & 7) { case 1: sp.groups = 2; break; case 2: sp.groups = 4; break; case 3: sp.groups = 16; break; case 4: sp.groups = 32; break; case 5: sp.groups = 8; break; default: break; }
    sp.unit_grad = unit_grad; sp.unit_max = unit_grad_max; sp.sums = sums; sp.gl_fwd = grad_loss_fwd; sp.gl_bwd = grad_loss_bwd;
    sp.gl_sum = grad_loss_sum; sp.gl_mean = grad_mean; sp.mean_div = (float)(batch_size / 2); sp.mean_of = mean_of;
    const int64_t blocks = (int64_t)batch_size * sp.groups;
    if (blocks > 0x7fffffffLL) return MR_ERR_BADARG;
    // (the covered-tile lists need the workgroup split's head room: grid > images; profiling bit 15 of flags: the listing form)
    if (scatter_work && sp.groups >= 2 && !(flags & (1 << 15))) {
        sp.work = scatter_work_at(const_cast<void*>(scatter_work), batch_size);
        hipLaunchKernelGGL(unit_scatter_tiles_kernel, dim3((unsigned)blocks), dim3(ST_WAVES * MR_WAVE), (size_t)table_bytes, s, sp);
    } else {
        hipLaunchKernelGGL(unit_scatter_listing_kernel, dim3((unsigned)blocks), dim3(ST_WAVES * MR_WAVE), (size_t)table_bytes, s, sp);
    }
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_flow_pair_backward_unit_tiles(const int32_t* face_index_map, const uint32_t* tile_hit,
                                                const float* weight_map, const int32_t* vertex_id_map, const float* unit_grad,
                                                const float* unit_grad_max, const float* sums, const float* grad_loss_fwd,
                                                const float* grad_loss_bwd, int height, int width, float* grad_vcolors,
                                                int batch_size, int num_verts, int num_faces, int fill_back, int image_size,
                                                float eps, int flags, int texel_layout, const void* scatter_work,
                                                mr_stream_t stream) {
    if (!grad_loss_fwd && (int64_t)batch_size * num_verts > 0) return MR_ERR_BADARG;
    return mr_flow_pair_backward_unit_tiles_ex(face_index_map, tile_hit, weight_map, vertex_id_map, unit_grad, unit_grad_max, sums,
                                               grad_loss_fwd, grad_loss_bwd, nullptr, nullptr, 0, height, width, grad_vcolors,
                                               batch_size, num_verts, num_faces, fill_back, image_size, eps, flags, texel_layout,
                                               scatter_work, stream);
}

extern "C" int mr_pixel_map_terms(uint64_t* terms_host, int reset) {
    if (!terms_host) return MR_ERR_BADARG;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    unsigned long long v = 0ull;
    e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(mr::mr_pixel_map_terms_counter), sizeof(v), 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    *terms_host = (uint64_t)v;
    if (reset) {
        v = 0ull;
        e = hipMemcpyToSymbol(HIP_SYMBOL(mr::mr_pixel_map_terms_counter), &v, sizeof(v), 0, hipMemcpyHostToDevice);
        if (e != hipSuccess) return (int)e;
    }
    return MR_OK;
}

#ifdef MR_WG_TIMELINE
extern "C" __attribute__((visibility("default"))) int mr_debug_ps_times(void* dst, long n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mr::mr_dbg_ps), n, 0, hipMemcpyDeviceToHost);
}
extern "C" __attribute__((visibility("default"))) int mr_debug_st_times(void* dst, long n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mr::mr_dbg_st), n, 0, hipMemcpyDeviceToHost);
}
#endif
