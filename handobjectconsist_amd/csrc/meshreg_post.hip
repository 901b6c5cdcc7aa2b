// meshreg_post.hip -- the parameter-free geometry between MeshRegNet's regression heads and the render path,
// for gfx950 (MI355X).
//
// Replaces, with their autograd (~280 small PyTorch launches per optimiser step of the trainer counterpart):
//   recover_3d_proj               meshreg/models/project.py:5-24   (weak-perspective camera recovery)
//   hand:   verts / joints in metres + recovered centre, joint projection      meshregnet.py:206-245
//   object: axis-angle -> rotation, rotated canonical vertices + recovered centre, projection
//                                                                              objbranch.py:28-84, meshregnet.py:274-323
//   libyana camutils.project.batch_proj2d  (K p)[:2] / (K p)[2]
// One thread per point (hand vertices, joints and object vertices of one sample share a block row); the
// per-sample quantities (two centres, one rotation) are recomputed by every thread from a dozen scalars.
// Backward: per-point adjoints, block-reduced per sample into 15 sums (two centres + d rotation), then one
// tiny kernel for the centre / Rodrigues adjoints.  3-term products are FMA chains in k order (as in the
// BLAS kernels behind torch.bmm).
#include "mr_common.hpp"

namespace mr {

struct PostParams {
    const float* verts_mm;    // [B,Vh,3]  MANO vertices, millimetres
    const float* joints_mm;   // [B,J,3]
    const float* scaletrans;  // [B,3]  (scale, tx, ty) of the hand
    const float* st_obj;      // [B,6]  (scale, tx, ty, axis-angle) of the object
    const float* K;           // [B,3,3]
    const float* canverts;    // [B,Vo,3]
    float trans_factor, scale_factor, off_z, res_w, res_h;
    int B, Vh, J, Vo;
};

// recovered centre (project.py:14-23): Z0 = f s + off_z ; XY0 = (t + img_centre - cam_centre) * Z0 / f
__device__ __forceinline__ void center3d(const float* K, float s, float tx, float ty, float off_z, float rw, float rh,
                                         float* c) {
    const float f = K[0];
    const float z0 = f * s + off_z;
    c[0] = (tx + rw / 2.0f - K[2]) * z0 / f;
    c[1] = (ty + rh / 2.0f - K[5]) * z0 / f;
    c[2] = z0;
}

__device__ __forceinline__ void proj2d_fwd(const float* K, const float* p, float* h, float* uv) {
#pragma unroll
    for (int i = 0; i < 3; i++) h[i] = fmaf(K[3 * i + 2], p[2], fmaf(K[3 * i + 1], p[1], K[3 * i] * p[0]));
    uv[0] = h[0] / h[2];
    uv[1] = h[1] / h[2];
}
// adjoint of proj2d_fwd: g_uv[2] -> += gp[3]
__device__ __forceinline__ void proj2d_bwd(const float* K, const float* h, const float* guv, float* gp) {
    const float gh[3] = {guv[0] / h[2], guv[1] / h[2], -(guv[0] * h[0] + guv[1] * h[1]) / (h[2] * h[2])};
#pragma unroll
    for (int j = 0; j < 3; j++) gp[j] += fmaf(K[6 + j], gh[2], fmaf(K[3 + j], gh[1], K[j] * gh[0]));
}

struct SampleGeo {
    float K[9], ch[3], co[3], R[9];
};
__device__ __forceinline__ SampleGeo sample_geo(const PostParams& p, int b) {
    SampleGeo g;
#pragma unroll
    for (int k = 0; k < 9; k++) g.K[k] = p.K[b * 9 + k];
    const float* st = p.scaletrans + b * 3;
    const float* so = p.st_obj + b * 6;
    center3d(g.K, st[0] * p.scale_factor, st[1] * p.trans_factor, st[2] * p.trans_factor, p.off_z, p.res_w, p.res_h, g.ch);
    center3d(g.K, so[0] * p.scale_factor, so[1] * p.trans_factor, so[2] * p.trans_factor, p.off_z, p.res_w, p.res_h, g.co);
    rodrigues(so + 3, g.R, nullptr);
    return g;
}

// grid (ceil(max(Vh + J, Vo) / 256), B)
__global__ void __launch_bounds__(256) post_forward_kernel(PostParams p, float* __restrict__ handverts3d,
                                                           float* __restrict__ joints3d, float* __restrict__ joints2d,
                                                           float* __restrict__ objverts3d,
                                                           float* __restrict__ objverts2d) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const SampleGeo g = sample_geo(p, b);
    if (i < p.Vh) {
        const int64_t o = ((int64_t)b * p.Vh + i) * 3;
#pragma unroll
        for (int r = 0; r < 3; r++) handverts3d[o + r] = p.verts_mm[o + r] / 1000.0f + g.ch[r];
    } else if (i < p.Vh + p.J) {
        const int64_t o = ((int64_t)b * p.J + (i - p.Vh)) * 3;
        float q[3], h[3], uv[2];
#pragma unroll
        for (int r = 0; r < 3; r++) { q[r] = g.ch[r] + p.joints_mm[o + r] / 1000.0f; joints3d[o + r] = q[r]; }
        proj2d_fwd(g.K, q, h, uv);
        joints2d[o / 3 * 2] = uv[0]; joints2d[o / 3 * 2 + 1] = uv[1];
    }
    if (i < p.Vo) {
        const int64_t o = ((int64_t)b * p.Vo + i) * 3;
        const float c[3] = {p.canverts[o], p.canverts[o + 1], p.canverts[o + 2]};
        float q[3], h[3], uv[2];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            q[r] = g.co[r] + fmaf(g.R[3 * r + 2], c[2], fmaf(g.R[3 * r + 1], c[1], g.R[3 * r] * c[0]));
            objverts3d[o + r] = q[r];
        }
        proj2d_fwd(g.K, q, h, uv);
        objverts2d[o / 3 * 2] = uv[0]; objverts2d[o / 3 * 2 + 1] = uv[1];
    }
}

constexpr int PS_NSUM = 15;  // d centre(hand) 3 | d centre(object) 3 | d R 9

// per-point adjoints + per-sample sums (block reduction, one float atomic per sum and block)
__global__ void __launch_bounds__(256) post_backward_kernel(PostParams p, const float* __restrict__ g_hv,
                                                            const float* __restrict__ g_j3,
                                                            const float* __restrict__ g_j2,
                                                            const float* __restrict__ g_ov,
                                                            const float* __restrict__ g_o2,
                                                            float* __restrict__ grad_verts_mm,
                                                            float* __restrict__ grad_joints_mm,
                                                            float* __restrict__ sums) {
    __shared__ float red[4][PS_NSUM];
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const SampleGeo g = sample_geo(p, b);
    float acc[PS_NSUM];
#pragma unroll
    for (int k = 0; k < PS_NSUM; k++) acc[k] = 0.0f;
    if (i < p.Vh) {
        const int64_t o = ((int64_t)b * p.Vh + i) * 3;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float gv = g_hv ? g_hv[o + r] : 0.0f;
            grad_verts_mm[o + r] = gv / 1000.0f;
            acc[r] += gv;
        }
    } else if (i < p.Vh + p.J) {
        const int64_t o = ((int64_t)b * p.J + (i - p.Vh)) * 3;
        float q[3], h[3], uv[2], gq[3];
#pragma unroll
        for (int r = 0; r < 3; r++) { q[r] = g.ch[r] + p.joints_mm[o + r] / 1000.0f; gq[r] = g_j3 ? g_j3[o + r] : 0.0f; }
        if (g_j2) {
            proj2d_fwd(g.K, q, h, uv);
            const float guv[2] = {g_j2[o / 3 * 2], g_j2[o / 3 * 2 + 1]};
            proj2d_bwd(g.K, h, guv, gq);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) { grad_joints_mm[o + r] = gq[r] / 1000.0f; acc[r] += gq[r]; }
    }
    if (i < p.Vo) {
        const int64_t o = ((int64_t)b * p.Vo + i) * 3;
        const float c[3] = {p.canverts[o], p.canverts[o + 1], p.canverts[o + 2]};
        float q[3], h[3], uv[2], gq[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            q[r] = g.co[r] + fmaf(g.R[3 * r + 2], c[2], fmaf(g.R[3 * r + 1], c[1], g.R[3 * r] * c[0]));
            gq[r] = g_ov ? g_ov[o + r] : 0.0f;
        }
        if (g_o2) {
            proj2d_fwd(g.K, q, h, uv);
            const float guv[2] = {g_o2[o / 3 * 2], g_o2[o / 3 * 2 + 1]};
            proj2d_bwd(g.K, h, guv, gq);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
            acc[3 + r] += gq[r];
#pragma unroll
            for (int k = 0; k < 3; k++) acc[6 + 3 * r + k] += gq[r] * c[k];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < PS_NSUM; k++) acc[k] += __shfl_xor(acc[k], off);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < PS_NSUM; k++) red[wave][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < PS_NSUM) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (v != 0.0f) atomicAdd(&sums[b * PS_NSUM + threadIdx.x], v);
    }
}

// adjoint of center3d: gc[3] -> (d s, d tx, d ty) of the raw head outputs (factors applied)
__device__ __forceinline__ void center3d_bwd(const float* K, float s_raw, float tx_raw, float ty_raw, const PostParams& p,
                                             const float* gc, float* out3) {
    const float f = K[0];
    const float s = s_raw * p.scale_factor, tx = tx_raw * p.trans_factor, ty = ty_raw * p.trans_factor;
    const float z0 = f * s + p.off_z;
    const float ax = tx + p.res_w / 2.0f - K[2], ay = ty + p.res_h / 2.0f - K[5];
    const float g_z0 = gc[2] + gc[0] * ax / f + gc[1] * ay / f;
    out3[0] = g_z0 * f * p.scale_factor;
    out3[1] = gc[0] * z0 / f * p.trans_factor;
    out3[2] = gc[1] * z0 / f * p.trans_factor;
}

__global__ void __launch_bounds__(64) post_backward_finish_kernel(PostParams p, const float* __restrict__ sums,
                                                                  float* __restrict__ grad_scaletrans,
                                                                  float* __restrict__ grad_st_obj) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    float K[9];
#pragma unroll
    for (int k = 0; k < 9; k++) K[k] = p.K[b * 9 + k];
    const float* st = p.scaletrans + b * 3;
    const float* so = p.st_obj + b * 6;
    const float* s = sums + b * PS_NSUM;
    float o3[3];
    center3d_bwd(K, st[0], st[1], st[2], p, s, o3);
#pragma unroll
    for (int k = 0; k < 3; k++) grad_scaletrans[b * 3 + k] = o3[k];
    center3d_bwd(K, so[0], so[1], so[2], p, s + 3, o3);
#pragma unroll
    for (int k = 0; k < 3; k++) grad_st_obj[b * 6 + k] = o3[k];
    float gr[3];
    rodrigues_bwd(so + 3, s + 6, gr);
#pragma unroll
    for (int k = 0; k < 3; k++) grad_st_obj[b * 6 + 3 + k] = gr[k];
}

}  // namespace mr

using namespace mr;

static int post_check(const PostParams& p) {
    if (p.B < 0 || p.Vh < 0 || p.J < 0 || p.Vo < 0) return MR_ERR_BADARG;
    if (p.B > 65535) return MR_ERR_BADARG;
    if (p.B == 0) return MR_OK;
    if (!p.scaletrans || !p.st_obj || !p.K) return MR_ERR_BADARG;
    if ((p.Vh > 0 && !p.verts_mm) || (p.J > 0 && !p.joints_mm) || (p.Vo > 0 && !p.canverts)) return MR_ERR_BADARG;
    return MR_OK;
}

extern "C" int mr_meshreg_post_forward(const float* verts_mm, const float* joints_mm, const float* scaletrans,
                                       const float* st_obj, const float* K, const float* canverts, float trans_factor,
                                       float scale_factor, float off_z, float res_w, float res_h, float* handverts3d,
                                       float* joints3d, float* joints2d, float* objverts3d, float* objverts2d,
                                       int batch_size, int num_hand_verts, int num_joints, int num_obj_verts,
                                       mr_stream_t stream) {
    PostParams p{verts_mm, joints_mm, scaletrans, st_obj, K, canverts, trans_factor, scale_factor, off_z, res_w, res_h,
                 batch_size, num_hand_verts, num_joints, num_obj_verts};
    const int rc = post_check(p);
    if (rc != MR_OK || batch_size == 0) return rc;
    if ((num_hand_verts > 0 && !handverts3d) || (num_joints > 0 && (!joints3d || !joints2d)) ||
        (num_obj_verts > 0 && (!objverts3d || !objverts2d)))
        return MR_ERR_BADARG;
    const int n = max(num_hand_verts + num_joints, num_obj_verts);
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(post_forward_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)batch_size), dim3(256), 0,
                       (hipStream_t)stream, p, handverts3d, joints3d, joints2d, objverts3d, objverts2d);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_meshreg_post_backward(const float* verts_mm, const float* joints_mm, const float* scaletrans,
                                        const float* st_obj, const float* K, const float* canverts, float trans_factor,
                                        float scale_factor, float off_z, float res_w, float res_h,
                                        const float* grad_handverts3d, const float* grad_joints3d,
                                        const float* grad_joints2d, const float* grad_objverts3d,
                                        const float* grad_objverts2d, float* workspace, float* grad_verts_mm,
                                        float* grad_joints_mm, float* grad_scaletrans, float* grad_st_obj,
                                        int batch_size, int num_hand_verts, int num_joints, int num_obj_verts,
                                        mr_stream_t stream) {
    PostParams p{verts_mm, joints_mm, scaletrans, st_obj, K, canverts, trans_factor, scale_factor, off_z, res_w, res_h,
                 batch_size, num_hand_verts, num_joints, num_obj_verts};
    const int rc = post_check(p);
    if (rc != MR_OK || batch_size == 0) return rc;
    if (!workspace || !grad_scaletrans || !grad_st_obj || (num_hand_verts > 0 && !grad_verts_mm) ||
        (num_joints > 0 && !grad_joints_mm))
        return MR_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)batch_size * PS_NSUM * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    const int n = max(num_hand_verts + num_joints, num_obj_verts);
    if (n > 0) {
        hipLaunchKernelGGL(post_backward_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)batch_size), dim3(256), 0, s, p,
                           grad_handverts3d, grad_joints3d, grad_joints2d, grad_objverts3d, grad_objverts2d,
                           grad_verts_mm, grad_joints_mm, workspace);
        MR_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(post_backward_finish_kernel, dim3((unsigned)((batch_size + 63) / 64)), dim3(64), 0, s, p,
                       (const float*)workspace, grad_scaletrans, grad_st_obj);
    MR_CHECK_LAUNCH();
    return MR_OK;
}
