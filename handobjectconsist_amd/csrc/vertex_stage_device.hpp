// vertex_stage_device.hpp -- the per-vertex arithmetic of the frame pair's front end (batch_proj2d, nr.projection) and the
// argument blocks of its launches, shared by vertex_stage.hip (the launches of their own) and raster_fwd.hip (round 6: the
// binning pass of a pair step computes its image's vertices itself, bin_boxes_kernel PROLOGUE).  ONE definition of the
// arithmetic: the two forms agree to the last bit by construction.
#pragma once
#include "mr_common.hpp"

namespace mr {

struct VertexStageParams {
    const float* verts1;  // [B,V,3] camera frame -- or, with part B given, the first `split` vertices [B,split,3]
    const float* verts2;
    const float* verts1b; // nullable: vertices split .. V - 1 of every mesh [B,V-split,3] (hand | object: the concatenation
    const float* verts2b; //           torch.cat([hand, obj], 1) of warpbranch.py:49-55 done by index instead of by a copy)
    int split;
    const float* K1;      // [B,3,3]
    const float* K2;
    const float* R;       // [Bc,3,3]  (Bc = 1 or B)
    const float* t;       // [Bc,3]
    const float* dist;    // [Bc,5]
    int cam_bstride;      // 0 (broadcast) or 1
    float orig_size;
    float* ndc1;          // [B,V,3]
    float* ndc2;
    float* cols12;        // [B,V,3] = (p2 - p1, 1)
    float* cols21;        // [B,V,3] = (p1 - p2, 1)
    int B, V;
};

__device__ __forceinline__ float dot3(const float* a, const float* b) {
    return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}

// batch_proj2d: (K v)[:2] / (K v)[2]
__device__ __forceinline__ void proj2d(const float* K, const float* v, float* h, float& px, float& py) {
#pragma unroll
    for (int i = 0; i < 3; i++) h[i] = dot3(K + 3 * i, v);
    px = h[0] / h[2];
    py = h[1] / h[2];
}

// nr.projection
__device__ __forceinline__ void ndc_project(const float* K, const float* R, const float* t, const float* d, float os,
                                            const float* v, float* out) {
    float c[3];
#pragma unroll
    for (int i = 0; i < 3; i++) c[i] = dot3(R + 3 * i, v) + t[i];
    const float z = c[2];
    const float x_ = c[0] / (z + 1e-9f), y_ = c[1] / (z + 1e-9f);
    const float k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4];
    const float r = sqrtf(x_ * x_ + y_ * y_);
    const float r2 = r * r, r4 = r2 * r2, r6 = r4 * r2;
    const float rad = 1.0f + k1 * r2 + k2 * r4 + k3 * r6;
    const float x__ = x_ * rad + 2.0f * p1 * x_ * y_ + p2 * (r2 + 2.0f * (x_ * x_));
    const float y__ = y_ * rad + p1 * (r2 + 2.0f * (y_ * y_)) + 2.0f * p2 * x_ * y_;
    const float xy1[3] = {x__, y__, 1.0f};
    float u = dot3(K, xy1);
    float w = dot3(K + 3, xy1);
    w = os - w;
    out[0] = 2.0f * (u - os / 2.0f) / os;
    out[1] = 2.0f * (w - os / 2.0f) / os;
    out[2] = z;
}

struct StackFacesParams {
    const int64_t* hand_faces;
    int64_t hand_bstride;
    const int64_t* obj_faces;
    int offset;
    int32_t* out;
    int B, Fh, Fo;
};
// What bin_boxes_kernel PROLOGUE (raster_fwd.hip) needs of the pair's front end: the workgroup(s) of stack image b compute
// the image's projected vertices into LDS (they exist nowhere else), write its flow colours and its rows of the stacked faces.
struct PairPrologue {
    VertexStageParams v;  // (ndc1 / ndc2 are not written)
    StackFacesParams f;
};

// the pair's set-up as a launch of its own, from its argument blocks (vertex_stage.hip; no clearing)
int mr_launch_pair_prologue(const PairPrologue& pro, hipStream_t s);

// the camera of pair pb in registers (wave-uniform)
struct PairCamera {
    float K1[9], K2[9], R[9], t[3], d[5];
};
__device__ __forceinline__ void load_pair_camera(const VertexStageParams& p, int pb, PairCamera& c) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
        c.K1[k] = p.K1[pb * 9 + k]; c.K2[k] = p.K2[pb * 9 + k];
        c.R[k] = p.R[(int64_t)pb * p.cam_bstride * 9 + k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) c.t[k] = p.t[(int64_t)pb * p.cam_bstride * 3 + k];
#pragma unroll
    for (int k = 0; k < 5; k++) c.d[k] = p.dist[(int64_t)pb * p.cam_bstride * 5 + k];
}

// vertex vi of pair pb as the stack image of `frame` (0: frame 1, 1: frame 2) needs it: ndc[3] = nr.projection of the frame's
// own vertex, col[2] = the displacement towards the other frame (p2 - p1 for frame 1, p1 - p2 for frame 2) -- the operations of
// flow_vertices_forward_body, which computes both frames' values in one thread
__device__ __forceinline__ void pair_vertex_of_frame(const VertexStageParams& p, const PairCamera& cam, int pb, int vi, int frame,
                                                     float* ndc, float* col) {
    const bool second = p.verts1b != nullptr && vi >= p.split;
    const float* s1 = second ? p.verts1b + ((int64_t)pb * (p.V - p.split) + (vi - p.split)) * 3
                             : p.verts1 + ((int64_t)pb * (p.verts1b ? p.split : p.V) + vi) * 3;
    const float* s2 = second ? p.verts2b + ((int64_t)pb * (p.V - p.split) + (vi - p.split)) * 3
                             : p.verts2 + ((int64_t)pb * (p.verts2b ? p.split : p.V) + vi) * 3;
    const float v1[3] = {s1[0], s1[1], s1[2]};
    const float v2[3] = {s2[0], s2[1], s2[2]};
    float h[3], a[2], c[2];
    proj2d(cam.K1, v1, h, a[0], a[1]);
    proj2d(cam.K2, v2, h, c[0], c[1]);
    if (frame == 0) {
        col[0] = c[0] - a[0]; col[1] = c[1] - a[1];
        ndc_project(cam.K1, cam.R, cam.t, cam.d, p.orig_size, v1, ndc);
    } else {
        col[0] = a[0] - c[0]; col[1] = a[1] - c[1];
        ndc_project(cam.K2, cam.R, cam.t, cam.d, p.orig_size, v2, ndc);
    }
}

// vertex index k (0..2) of face f of the pair's concatenated mesh (stack_pair_faces_body's value for element 3 f + k)
__device__ __forceinline__ int32_t pair_face_index(const StackFacesParams& q, int pb, int f, int k) {
    const int i = f * 3 + k, f3h = q.Fh * 3;
    const int64_t v = i < f3h ? q.hand_faces[(int64_t)pb * q.hand_bstride + i] : q.obj_faces[(int64_t)pb * q.Fo * 3 + (i - f3h)] + q.offset;
    return (int32_t)v;
}

}  // namespace mr
