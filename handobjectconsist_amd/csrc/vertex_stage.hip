// vertex_stage.hip -- the per-vertex front end of get_opticalflow for gfx950 (SURVEY 8f "f1").
//
// Replaces, for the training setting (vertex-colour render, detach_renders=True):
//   libyana camutils.project.batch_proj2d on both frames           (opticalflow.py:98-99)
//   the two displacement textures  (p2 - p1, 1) and (p1 - p2, 1)    (opticalflow.py:101-102, 121-122)
//   Renderer.project -> nr.projection of both frames               (renderer.py:164-188, 278)
// and their autograd, ~150 small element-wise / bmm launches over [B,V,3] tensors, by ONE kernel
// per direction.  One thread per (image, vertex); the camera matrices of the image are wave-uniform.
// Arithmetic follows the operation order of the PyTorch expressions it replaces; the 3-term products of
// the matmuls are FMA chains in k order (what the BLAS kernels behind torch.bmm / numpy.matmul do), so the
// projected vertices -- and with them every coverage / validity decision downstream -- agree with the
// op-by-op path to the last bit in practice.
#include "mr_common.hpp"
#include "vertex_stage_device.hpp"

namespace mr {

__device__ __forceinline__ void flow_vertices_forward_body(const VertexStageParams& p, int bx) {
    const int b = blockIdx.y;
    const int vi = bx * blockDim.x + threadIdx.x;
    if (vi >= p.V) return;
    float K1[9], K2[9], R[9], t[3], d[5];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        K1[k] = p.K1[b * 9 + k]; K2[k] = p.K2[b * 9 + k];
        R[k] = p.R[(int64_t)b * p.cam_bstride * 9 + k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = p.t[(int64_t)b * p.cam_bstride * 3 + k];
#pragma unroll
    for (int k = 0; k < 5; k++) d[k] = p.dist[(int64_t)b * p.cam_bstride * 5 + k];
    const int64_t o = ((int64_t)b * p.V + vi) * 3;
    const bool second = p.verts1b != nullptr && vi >= p.split;
    const float* s1 = second ? p.verts1b + ((int64_t)b * (p.V - p.split) + (vi - p.split)) * 3
                             : p.verts1 + ((int64_t)b * (p.verts1b ? p.split : p.V) + vi) * 3;
    const float* s2 = second ? p.verts2b + ((int64_t)b * (p.V - p.split) + (vi - p.split)) * 3
                             : p.verts2 + ((int64_t)b * (p.verts2b ? p.split : p.V) + vi) * 3;
    const float v1[3] = {s1[0], s1[1], s1[2]};
    const float v2[3] = {s2[0], s2[1], s2[2]};
    float h[3], a[2], c[2];
    proj2d(K1, v1, h, a[0], a[1]);
    proj2d(K2, v2, h, c[0], c[1]);
    p.cols12[o] = c[0] - a[0]; p.cols12[o + 1] = c[1] - a[1]; p.cols12[o + 2] = 1.0f;
    p.cols21[o] = a[0] - c[0]; p.cols21[o + 1] = a[1] - c[1]; p.cols21[o + 2] = 1.0f;
    float n[3];
    ndc_project(K1, R, t, d, p.orig_size, v1, n);
    p.ndc1[o] = n[0]; p.ndc1[o + 1] = n[1]; p.ndc1[o + 2] = n[2];
    ndc_project(K2, R, t, d, p.orig_size, v2, n);
    p.ndc2[o] = n[0]; p.ndc2[o + 1] = n[1]; p.ndc2[o + 2] = n[2];
}

__global__ void __launch_bounds__(256) flow_vertices_forward_kernel(VertexStageParams p) { flow_vertices_forward_body(p, blockIdx.x); }

// adjoint of (verts1, verts2) -> (cols12, cols21); grad_verts* may be NULL
// (two-part meshes: verts*b / grad_verts*b hold the vertices from `split` on; a NULL gradient pointer = not wanted)
__global__ void __launch_bounds__(256) flow_vertices_backward_kernel(const float* __restrict__ verts1,
                                                                     const float* __restrict__ verts2,
                                                                     const float* __restrict__ K1g,
                                                                     const float* __restrict__ K2g,
                                                                     const float* __restrict__ g12,
                                                                     const float* __restrict__ g21,
                                                                     float* __restrict__ grad_verts1,
                                                                     float* __restrict__ grad_verts2, int B, int V,
                                                                     const float* __restrict__ verts1b,
                                                                     const float* __restrict__ verts2b,
                                                                     float* __restrict__ grad_verts1b,
                                                                     float* __restrict__ grad_verts2b, int split) {
    const int b = blockIdx.y;
    const int vi = blockIdx.x * blockDim.x + threadIdx.x;
    if (vi >= V) return;
    const int64_t o = ((int64_t)b * V + vi) * 3;
    const bool parts = verts1b != nullptr, second = parts && vi >= split;
    const int64_t op = second ? ((int64_t)b * (V - split) + (vi - split)) * 3 : ((int64_t)b * (parts ? split : V) + vi) * 3;
    // d loss / d p1 = g21 - g12,  d loss / d p2 = g12 - g21  (x, y components; the constant 1 has no gradient)
    float ga[2] = {0.0f, 0.0f};
    if (g12) { ga[0] += g12[o]; ga[1] += g12[o + 1]; }
    if (g21) { ga[0] -= g21[o]; ga[1] -= g21[o + 1]; }
#pragma unroll
    for (int f = 0; f < 2; f++) {
        float* out = f == 0 ? (second ? grad_verts1b : grad_verts1) : (second ? grad_verts2b : grad_verts2);
        if (!out) continue;
        const float* K = (f == 0 ? K1g : K2g) + b * 9;
        const float* vp = (f == 0 ? (second ? verts1b : verts1) : (second ? verts2b : verts2)) + op;
        const float v[3] = {vp[0], vp[1], vp[2]};
        const float sgn = f == 0 ? -1.0f : 1.0f;  // p1 enters cols12 with a minus sign
        const float gp[2] = {sgn * ga[0], sgn * ga[1]};
        float h[3];
#pragma unroll
        for (int i = 0; i < 3; i++) h[i] = K[3 * i] * v[0] + K[3 * i + 1] * v[1] + K[3 * i + 2] * v[2];
        // p = (h0 / h2, h1 / h2)
        const float gh[3] = {gp[0] / h[2], gp[1] / h[2], -(gp[0] * h[0] + gp[1] * h[1]) / (h[2] * h[2])};
#pragma unroll
        for (int j = 0; j < 3; j++) out[op + j] = K[j] * gh[0] + K[3 + j] * gh[1] + K[6 + j] * gh[2];
    }
}

// faces of the concatenated hand + object mesh of a frame pair, as the stacked render takes them: int32 [2B, Fh + Fo, 3],
// rows [0, B) and [B, 2B) identical (both frames of a pair share the faces, warpbranch.py:49-55), object indices offset
// by the hand's vertex count -- hand_face.repeat + (obj_faces + Vh) + cat + cat + dtype conversion in one pass
__device__ __forceinline__ void stack_pair_faces_body(const StackFacesParams& q, int bx) {
    const int b = blockIdx.y;
    const int i = bx * blockDim.x + threadIdx.x;
    const int n = (q.Fh + q.Fo) * 3;
    if (i >= n) return;
    const int f3h = q.Fh * 3;
    const int64_t v = i < f3h ? q.hand_faces[(int64_t)b * q.hand_bstride + i] : q.obj_faces[(int64_t)b * q.Fo * 3 + (i - f3h)] + q.offset;
    q.out[(int64_t)b * n + i] = (int32_t)v;
    q.out[((int64_t)q.B + b) * n + i] = (int32_t)v;
}
__global__ void __launch_bounds__(256) stack_pair_faces_kernel(StackFacesParams q) { stack_pair_faces_body(q, blockIdx.x); }

// both set-up steps of a frame pair in ONE launch (mr_flow_pair_prologue_parts): the first `vertex_blocks` workgroups of a
// row project vertices, the others stack faces -- the two do not depend on each other
__global__ void __launch_bounds__(256) pair_prologue_kernel(VertexStageParams p, StackFacesParams q, int vertex_blocks,
                                                            uint4* __restrict__ clear16, int clear_words) {
    // (the header of the render's tile list and the arrival counters behind it, for MR_FLAG_TILE_LIST_CLEARED: the render
    // that follows on this stream adds to them)
    if (clear16 && blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < clear_words; i += blockDim.x) clear16[i] = make_uint4(0u, 0u, 0u, 0u);
    if ((int)blockIdx.x < vertex_blocks) flow_vertices_forward_body(p, blockIdx.x);
    else stack_pair_faces_body(q, (int)blockIdx.x - vertex_blocks);
}

}  // namespace mr

using namespace mr;

// the pair's set-up launch from its argument blocks (raster_fwd.hip's launch_bins: where the binning pass cannot take the
// vertex stage in); no clearing
int mr::mr_launch_pair_prologue(const mr::PairPrologue& pro, hipStream_t s) {
    const int vb = (pro.v.V + 255) / 256, fb = ((pro.f.Fh + pro.f.Fo) * 3 + 255) / 256;
    if (pro.v.B <= 0 || pro.v.B > 65535 || !pro.v.ndc1 || !pro.v.ndc2 || !pro.v.cols12 || !pro.v.cols21) return MR_ERR_BADARG;
    hipLaunchKernelGGL(pair_prologue_kernel, dim3((unsigned)(vb + fb), (unsigned)pro.v.B), dim3(256), 0, s, pro.v, pro.f, vb,
                       (uint4*)nullptr, 0);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_flow_vertices_forward(const float* verts1, const float* verts2, const float* K1, const float* K2,
                                        const float* R, const float* t, const float* dist_coeffs, int cam_batched,
                                        float orig_size, float* ndc1, float* ndc2, float* cols12, float* cols21,
                                        int batch_size, int num_verts, mr_stream_t stream) {
    if (batch_size < 0 || num_verts < 0 || !(orig_size > 0.0f)) return MR_ERR_BADARG;
    if (batch_size == 0 || num_verts == 0) return MR_OK;
    if (!verts1 || !verts2 || !K1 || !K2 || !R || !t || !dist_coeffs || !ndc1 || !ndc2 || !cols12 || !cols21)
        return MR_ERR_BADARG;
    if (batch_size > 65535) return MR_ERR_BADARG;
    VertexStageParams p{verts1, verts2, nullptr, nullptr, num_verts, K1, K2, R, t, dist_coeffs, cam_batched ? 1 : 0, orig_size,
                        ndc1, ndc2, cols12, cols21, batch_size, num_verts};
    hipLaunchKernelGGL(flow_vertices_forward_kernel, dim3((unsigned)((num_verts + 255) / 256), (unsigned)batch_size),
                       dim3(256), 0, (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_flow_vertices_backward(const float* verts1, const float* verts2, const float* K1, const float* K2,
                                         const float* grad_cols12, const float* grad_cols21, float* grad_verts1,
                                         float* grad_verts2, int batch_size, int num_verts, mr_stream_t stream) {
    if (batch_size < 0 || num_verts < 0) return MR_ERR_BADARG;
    if (batch_size == 0 || num_verts == 0 || (!grad_verts1 && !grad_verts2)) return MR_OK;
    if (!verts1 || !verts2 || !K1 || !K2) return MR_ERR_BADARG;
    if (batch_size > 65535) return MR_ERR_BADARG;
    hipLaunchKernelGGL(flow_vertices_backward_kernel, dim3((unsigned)((num_verts + 255) / 256), (unsigned)batch_size),
                       dim3(256), 0, (hipStream_t)stream, verts1, verts2, K1, K2, grad_cols12, grad_cols21, grad_verts1,
                       grad_verts2, batch_size, num_verts, (const float*)nullptr, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr, num_verts);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_flow_vertices_parts_forward(const float* verts1a, const float* verts1b, const float* verts2a,
                                              const float* verts2b, int num_verts_a, int num_verts_b, const float* K1,
                                              const float* K2, const float* R, const float* t, const float* dist_coeffs,
                                              int cam_batched, float orig_size, float* ndc1, float* ndc2, float* cols12,
                                              float* cols21, int batch_size, mr_stream_t stream) {
    if (batch_size < 0 || num_verts_a < 0 || num_verts_b <= 0 || !(orig_size > 0.0f)) return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    if (!verts1b || !verts2b || ((!verts1a || !verts2a) && num_verts_a > 0) || !K1 || !K2 || !R || !t || !dist_coeffs ||
        !ndc1 || !ndc2 || !cols12 || !cols21)
        return MR_ERR_BADARG;
    if (batch_size > 65535) return MR_ERR_BADARG;
    const int V = num_verts_a + num_verts_b;
    VertexStageParams p{verts1a, verts2a, verts1b, verts2b, num_verts_a, K1, K2, R, t, dist_coeffs, cam_batched ? 1 : 0,
                        orig_size, ndc1, ndc2, cols12, cols21, batch_size, V};
    hipLaunchKernelGGL(flow_vertices_forward_kernel, dim3((unsigned)((V + 255) / 256), (unsigned)batch_size), dim3(256), 0,
                       (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_flow_vertices_parts_backward(const float* verts1a, const float* verts1b, const float* verts2a,
                                               const float* verts2b, int num_verts_a, int num_verts_b, const float* K1,
                                               const float* K2, const float* grad_cols12, const float* grad_cols21,
                                               float* grad_verts1a, float* grad_verts1b, float* grad_verts2a,
                                               float* grad_verts2b, int batch_size, mr_stream_t stream) {
    if (batch_size < 0 || num_verts_a < 0 || num_verts_b <= 0) return MR_ERR_BADARG;
    if (batch_size == 0 || (!grad_verts1a && !grad_verts1b && !grad_verts2a && !grad_verts2b)) return MR_OK;
    if (!verts1b || !verts2b || ((!verts1a || !verts2a) && num_verts_a > 0) || !K1 || !K2) return MR_ERR_BADARG;
    if (batch_size > 65535) return MR_ERR_BADARG;
    const int V = num_verts_a + num_verts_b;
    hipLaunchKernelGGL(flow_vertices_backward_kernel, dim3((unsigned)((V + 255) / 256), (unsigned)batch_size), dim3(256), 0,
                       (hipStream_t)stream, verts1a, verts2a, K1, K2, grad_cols12, grad_cols21, grad_verts1a, grad_verts2a,
                       batch_size, V, verts1b, verts2b, grad_verts1b, grad_verts2b, num_verts_a);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_stack_pair_faces(const int64_t* hand_faces, int hand_batched, const int64_t* obj_faces, int vertex_offset,
                                   int32_t* faces_out, int batch_size, int num_hand_faces, int num_obj_faces,
                                   mr_stream_t stream) {
    if (batch_size < 0 || num_hand_faces < 0 || num_obj_faces < 0 || vertex_offset < 0) return MR_ERR_BADARG;
    const int n = (num_hand_faces + num_obj_faces) * 3;
    if (batch_size == 0 || n == 0) return MR_OK;
    if (!faces_out || (!hand_faces && num_hand_faces > 0) || (!obj_faces && num_obj_faces > 0)) return MR_ERR_BADARG;
    if (batch_size > 65535) return MR_ERR_BADARG;
    const StackFacesParams q{hand_faces, hand_batched ? (int64_t)num_hand_faces * 3 : (int64_t)0, obj_faces, vertex_offset, faces_out,
                             batch_size, num_hand_faces, num_obj_faces};
    hipLaunchKernelGGL(stack_pair_faces_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)batch_size), dim3(256), 0,
                       (hipStream_t)stream, q);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_flow_pair_prologue_parts(const float* verts1a, const float* verts1b, const float* verts2a, const float* verts2b,
                                           int num_verts_a, int num_verts_b, const float* K1, const float* K2, const float* R,
                                           const float* t, const float* dist_coeffs, int cam_batched, float orig_size,
                                           float* ndc1, float* ndc2, float* cols12, float* cols21, const int64_t* hand_faces,
                                           int hand_batched, const int64_t* obj_faces, int32_t* faces_out, int num_hand_faces,
                                           int num_obj_faces, int batch_size, void* clear16, int64_t clear_bytes,
                                           mr_stream_t stream) {
    if (batch_size < 0 || num_verts_a < 0 || num_verts_b <= 0 || !(orig_size > 0.0f) || num_hand_faces < 0 || num_obj_faces < 0)
        return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    if (!verts1b || !verts2b || ((!verts1a || !verts2a) && num_verts_a > 0) || !K1 || !K2 || !R || !t || !dist_coeffs ||
        !ndc1 || !ndc2 || !cols12 || !cols21)
        return MR_ERR_BADARG;
    const int n = (num_hand_faces + num_obj_faces) * 3;
    if (n > 0 && (!faces_out || (!hand_faces && num_hand_faces > 0) || (!obj_faces && num_obj_faces > 0))) return MR_ERR_BADARG;
    if (batch_size > 65535) return MR_ERR_BADARG;
    const int V = num_verts_a + num_verts_b;
    const VertexStageParams p{verts1a, verts2a, verts1b, verts2b, num_verts_a, K1, K2, R, t, dist_coeffs, cam_batched ? 1 : 0,
                              orig_size, ndc1, ndc2, cols12, cols21, batch_size, V};
    const StackFacesParams q{hand_faces, hand_batched ? (int64_t)num_hand_faces * 3 : (int64_t)0, obj_faces, num_verts_a, faces_out,
                             batch_size, num_hand_faces, num_obj_faces};
    const int vb = (V + 255) / 256, fb = (n + 255) / 256;
    if (((uintptr_t)clear16 & 15u) != 0 || clear_bytes < 0 || (clear_bytes & 15) != 0 || clear_bytes > (1 << 24)) return MR_ERR_BADARG;
    hipLaunchKernelGGL(pair_prologue_kernel, dim3((unsigned)(vb + fb), (unsigned)batch_size), dim3(256), 0, (hipStream_t)stream,
                       p, q, vb, clear_bytes > 0 ? (uint4*)clear16 : nullptr, (int)(clear_bytes / 16));
    MR_CHECK_LAUNCH();
    return MR_OK;
}
