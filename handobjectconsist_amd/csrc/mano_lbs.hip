// mano_lbs.hip -- MANO linear-blend skinning for gfx950 (MI355X): SURVEY 8a row a19 / 8f "f1".
//
// Replaces manopth's ManoLayer.forward (third-party; called at /root/reference/meshreg/models/
// manobranch.py:130-136; arithmetic as restated in SURVEY appendix B.10) and its autograd:
//   pose PCA -> axis-angle -> rotations (via quaternions) -> pose blend-shape features,
//   v_posed = template + [betas | pose features] x blend shapes            <- the one GEMM-shaped step
//   joints  = J_regressor x (template + shape blend shapes)  (pre-multiplied regressors)
//   kinematic chain of 16 rigid transforms, skinning, joints + finger tips, centring, mm scale.
// Kernels:
//   mano_pre_kernel       one wave per sample: PCA, rotations, joints, chain, corrected transforms
//   mano_blend_kernel     v_posed = coeff[B,K] x blend[K,2334] on the MATRIX CORES: v_mfma_f32_32x32x2_f32
//                         (fp32 in / fp32 accumulate -- bitwise a k-ordered fmaf chain, i.e. what the BLAS
//                         kernel it replaces computes), one 32 x 32 tile per wave
//   mano_skin_kernel      one thread per (sample, vertex): blended 3x4 transform, vertex, finger tips
//   mano_skin_bwd_kernel  per (sample, 256-vertex chunk): d v_posed and the chunk's share of d transforms
//   mano_blend_bwd_kernel d coeff = d v_posed[B,2334] x blend^T on the matrix cores, split-K partials
//   mano_pre_bwd_kernel   one wave per sample: sums the partials, chain / Rodrigues / PCA adjoints
// This is the only GEMM-shaped work on the whole path (M = 2334, K = 145, N = batch).
#include "mr_common.hpp"

namespace mr {

constexpr int MN_V = 778, MN_J = 16, MN_NV3 = MN_V * 3, MN_KP = 146;  // K = 10 + 135, padded to even
constexpr int MN_TIPS = 5, MN_JT = MN_J + MN_TIPS;
constexpr int MN_SPLITK = 32;  // split-K factor of the backward GEMM

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ManoConst {
    const float* comps;    // [ncomps,45]
    const float* mean;     // [45]
    const float* js;       // [48,10]   J_regressor x shapedirs
    const float* jt;       // [48]      J_regressor x template
    const float* blend;    // [MN_KP, 2334]  k-major: rows 0..9 shapedirs, 10..144 posedirs, 145 zero
    const float* templ;    // [2334]
    const float* weights;  // [778,16]
    const int* parents;    // [16], parents[0] = -1
    const int* tips;       // [5] vertex ids
    const int* reorder;    // [21] output joint k = cat(joints, tips)[reorder[k]]
    int ncomps, center;    // center: index into cat(joints, tips) BEFORE the reorder, < 16 (a joint), or -1
};

// ---------------------------------------------------------------------------------------------------
// small helpers (per-lane serial 3x3 / 3x4 algebra; everything here is a few hundred flops per sample)
// ---------------------------------------------------------------------------------------------------
// 3x4 rigid transforms [R | t], row-major 12 floats: C = A o B
__device__ __forceinline__ void rigid_mul(const float* A, const float* B, float* C) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float v = fmaf(A[4 * r + 2], B[8 + c], fmaf(A[4 * r + 1], B[4 + c], A[4 * r] * B[c]));
            if (c == 3) v += A[4 * r + 3];
            C[4 * r + c] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// one wave per sample.  Outputs: coeff[B,MN_KP], G[B,16,12] (chain), G2[B,16,12] (rest-pose corrected),
// rots[B,16,9], joints[B,16,3]; the 16 joint rows of jtr_out (centred, x1000, reordered).
__global__ void __launch_bounds__(64) mano_pre_kernel(ManoConst mc, const float* __restrict__ pose,
                                                      const float* __restrict__ betas, float* __restrict__ coeff,
                                                      float* __restrict__ G, float* __restrict__ G2,
                                                      float* __restrict__ rots, float* __restrict__ joints,
                                                      float* __restrict__ full_pose_out, float* __restrict__ jtr_out,
                                                      int B) {
    __shared__ float s_pose[48], s_R[MN_J * 9], s_J[48], s_G[MN_J * 12];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int np = 3 + mc.ncomps;
    if (lane < 3) s_pose[lane] = pose[b * np + lane];
    if (lane < 45) {
        float acc = 0.0f;
        for (int c = 0; c < mc.ncomps; c++) acc = fmaf(pose[b * np + 3 + c], mc.comps[c * 45 + lane], acc);
        s_pose[3 + lane] = mc.mean[lane] + acc;
    }
    if (lane < 48) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 10; k++) acc = fmaf(betas[b * 10 + k], mc.js[lane * 10 + k], acc);
        s_J[lane] = acc + mc.jt[lane];
    }
    __syncthreads();
    if (lane < 48) {
        full_pose_out[b * 48 + lane] = s_pose[lane];
        joints[b * 48 + lane] = s_J[lane];
    }
    if (lane < MN_J) {
        float R[9];
        rodrigues(&s_pose[3 * lane], R, nullptr);
#pragma unroll
        for (int k = 0; k < 9; k++) {
            s_R[lane * 9 + k] = R[k];
            rots[(b * MN_J + lane) * 9 + k] = R[k];
            if (lane > 0) coeff[b * MN_KP + 10 + (lane - 1) * 9 + k] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f);
        }
    }
    if (lane < 10) coeff[b * MN_KP + lane] = betas[b * 10 + lane];
    if (lane == 10) coeff[b * MN_KP + MN_KP - 1] = 0.0f;
    __syncthreads();
    if (lane == 0) {  // the chain: 16 rigid products, a few hundred flops
        for (int j = 0; j < MN_J; j++) {
            const int pa = mc.parents[j];
            float rel[12];
#pragma unroll
            for (int r = 0; r < 3; r++) {
#pragma unroll
                for (int c = 0; c < 3; c++) rel[4 * r + c] = s_R[j * 9 + 3 * r + c];
                rel[4 * r + 3] = pa < 0 ? s_J[3 * j + r] : s_J[3 * j + r] - s_J[3 * pa + r];
            }
            if (pa < 0) {
#pragma unroll
                for (int k = 0; k < 12; k++) s_G[j * 12 + k] = rel[k];
            } else {
                rigid_mul(&s_G[pa * 12], rel, &s_G[j * 12]);
            }
        }
    }
    __syncthreads();
    for (int k = lane; k < MN_J * 12; k += 64) {
        const int j = k / 12, e = k % 12, r = e / 4, c = e % 4;
        float v = s_G[k];
        G[b * MN_J * 12 + k] = v;
        if (c == 3)  // t - R J
            v = v - fmaf(s_G[j * 12 + 4 * r + 2], s_J[3 * j + 2],
                         fmaf(s_G[j * 12 + 4 * r + 1], s_J[3 * j + 1], s_G[j * 12 + 4 * r] * s_J[3 * j]));
        G2[b * MN_J * 12 + k] = v;
    }
    // joint rows of the output (tips come from mano_skin_kernel)
    if (lane < MN_JT * 3) {
        const int k = lane / 3, r = lane % 3, src = mc.reorder[k];
        if (src < MN_J) {
            const float ctr = mc.center >= 0 ? s_G[mc.center * 12 + 4 * r + 3] : 0.0f;
            jtr_out[(b * MN_JT + k) * 3 + r] = (s_G[src * 12 + 4 * r + 3] - ctr) * 1000.0f;
        }
    }
}

// D[M = batch rows, N = 2334] = coeff[B, MN_KP] x blend[MN_KP, 2334] + template; one 32 x 32 tile per wave
__global__ void __launch_bounds__(64) mano_blend_kernel(const float* __restrict__ coeff, const float* __restrict__ blend,
                                                        const float* __restrict__ templ, float* __restrict__ v_posed,
                                                        int B) {
    const int lane = threadIdx.x;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int i = lane & 31, kh = lane >> 5;
    const int row = min(m0 + i, B - 1), col = min(n0 + i, MN_NV3 - 1);
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (int k = 0; k < MN_KP; k += 2) {
        const float a = coeff[row * MN_KP + k + kh];        // A[i][k]
        const float bb = blend[(k + kh) * MN_NV3 + col];    // B[k][j]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int n = n0 + i;
    if (n < MN_NV3) {
        const float t = templ[n];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m < B) v_posed[(int64_t)m * MN_NV3 + n] = acc[r] + t;
        }
    }
}

// one thread per (sample, vertex)
__global__ void __launch_bounds__(256) mano_skin_kernel(ManoConst mc, const float* __restrict__ v_posed,
                                                        const float* __restrict__ G, const float* __restrict__ G2,
                                                        float* __restrict__ verts_out, float* __restrict__ jtr_out,
                                                        int B) {
    __shared__ float s_G2[MN_J * 12];
    const int b = blockIdx.y, v = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = threadIdx.x; k < MN_J * 12; k += blockDim.x) s_G2[k] = G2[b * MN_J * 12 + k];
    __syncthreads();
    if (v >= MN_V) return;
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; e++) T[e] = 0.0f;
    for (int j = 0; j < MN_J; j++) {
        const float w = mc.weights[v * MN_J + j];
#pragma unroll
        for (int e = 0; e < 12; e++) T[e] = fmaf(w, s_G2[j * 12 + e], T[e]);
    }
    const float* vp = v_posed + ((int64_t)b * MN_V + v) * 3;
    const float p[3] = {vp[0], vp[1], vp[2]};
    float o[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        o[r] = fmaf(T[4 * r + 2], p[2], fmaf(T[4 * r + 1], p[1], T[4 * r] * p[0])) + T[4 * r + 3];
        const float ctr = mc.center >= 0 ? G[(b * MN_J + mc.center) * 12 + 4 * r + 3] : 0.0f;
        o[r] = (o[r] - ctr) * 1000.0f;
        verts_out[((int64_t)b * MN_V + v) * 3 + r] = o[r];
    }
#pragma unroll
    for (int tpi = 0; tpi < MN_TIPS; tpi++)
        if (mc.tips[tpi] == v) {
            for (int k = 0; k < MN_JT; k++)
                if (mc.reorder[k] == MN_J + tpi)
#pragma unroll
                    for (int r = 0; r < 3; r++) jtr_out[(b * MN_JT + k) * 3 + r] = o[r];
        }
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
// per (sample, chunk of 256 vertices): gradient of v_posed, and the chunk's share of dG2[16,12] and of the
// gradient of the centre (3), written to part[b][chunk][16 * 12 + 3]
constexpr int MN_CHUNKS = (MN_V + 255) / 256;
constexpr int MN_PART = MN_J * 12 + 3;

__global__ void __launch_bounds__(256) mano_skin_bwd_kernel(ManoConst mc, const float* __restrict__ v_posed,
                                                            const float* __restrict__ G2,
                                                            const float* __restrict__ grad_verts,
                                                            const float* __restrict__ grad_jtr,
                                                            float* __restrict__ grad_vp, float* __restrict__ part,
                                                            int B) {
    __shared__ float s_G2[MN_J * 12];
    __shared__ float s_gt[256][13];   // g (x) [vp, 1] per vertex of the chunk (12) -- padded
    __shared__ float s_w[256][MN_J + 1];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, v = chunk * 256 + tid;
    for (int k = tid; k < MN_J * 12; k += 256) s_G2[k] = G2[b * MN_J * 12 + k];
    __syncthreads();
    float g[3] = {0.0f, 0.0f, 0.0f};
    if (v < MN_V) {
        float T[12], w[MN_J];
#pragma unroll
        for (int e = 0; e < 12; e++) T[e] = 0.0f;
        for (int j = 0; j < MN_J; j++) {
            w[j] = mc.weights[v * MN_J + j];
            s_w[tid][j] = w[j];
#pragma unroll
            for (int e = 0; e < 12; e++) T[e] = fmaf(w[j], s_G2[j * 12 + e], T[e]);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) g[r] = grad_verts ? grad_verts[((int64_t)b * MN_V + v) * 3 + r] * 1000.0f : 0.0f;
        if (grad_jtr)
            for (int tpi = 0; tpi < MN_TIPS; tpi++)
                if (mc.tips[tpi] == v)
                    for (int k = 0; k < MN_JT; k++)
                        if (mc.reorder[k] == MN_J + tpi)
#pragma unroll
                            for (int r = 0; r < 3; r++) g[r] += grad_jtr[(b * MN_JT + k) * 3 + r] * 1000.0f;
        const float* vp = v_posed + ((int64_t)b * MN_V + v) * 3;
        const float p[4] = {vp[0], vp[1], vp[2], 1.0f};
#pragma unroll
        for (int c = 0; c < 3; c++)
            grad_vp[((int64_t)b * MN_V + v) * 3 + c] = fmaf(T[8 + c], g[2], fmaf(T[4 + c], g[1], T[c] * g[0]));
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) s_gt[tid][4 * r + c] = g[r] * p[c];
    } else {
#pragma unroll
        for (int e = 0; e < 12; e++) s_gt[tid][e] = 0.0f;
        for (int j = 0; j < MN_J; j++) s_w[tid][j] = 0.0f;
    }
    s_gt[tid][12] = 0.0f;
    __syncthreads();
    // dG2[j][e] = sum_v w[v][j] * gT[v][e] : one thread per (j, e), fixed order -> deterministic
    if (tid < MN_J * 12) {
        const int j = tid / 12, e = tid % 12;
        float acc = 0.0f;
        for (int u = 0; u < 256; u++) acc = fmaf(s_w[u][j], s_gt[u][e], acc);
        part[((int64_t)b * MN_CHUNKS + chunk) * MN_PART + tid] = acc;
    } else if (tid < MN_J * 12 + 3) {
        // d centre = - sum of the vertex gradients (each vertex subtracts the centre); g = gT[.][4 r + 3]
        const int r = tid - MN_J * 12;
        float acc = 0.0f;
        for (int u = 0; u < 256; u++) acc += s_gt[u][4 * r + 3];
        part[((int64_t)b * MN_CHUNKS + chunk) * MN_PART + tid] = -acc;
    }
}

// grad_coeff partial[s][B, 160] = grad_vp[B, 2334 (slice s)] x blend^T: split-K on the matrix cores
__global__ void __launch_bounds__(64) mano_blend_bwd_kernel(const float* __restrict__ grad_vp,
                                                            const float* __restrict__ blend,
                                                            float* __restrict__ partial, int B, int Mpad) {
    const int lane = threadIdx.x;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, s = blockIdx.z;
    const int i = lane & 31, kh = lane >> 5;
    const int row = min(m0 + i, B - 1), col = min(n0 + i, MN_KP - 1);
    const int kper = ((MN_NV3 + MN_SPLITK - 1) / MN_SPLITK + 1) & ~1;  // even slice length
    const int k0 = s * kper, k1 = min(k0 + kper, MN_NV3);
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = k0; k < k1; k += 2) {
        const int kk = k + kh;
        const bool in = kk < k1;
        const float a = in ? grad_vp[(int64_t)row * MN_NV3 + kk] : 0.0f;   // A[i][k]
        const float bb = in ? blend[col * MN_NV3 + kk] : 0.0f;            // B[k][j] = blend[j][k]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
    }
    const int n = n0 + i;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (m < Mpad && n < 160) partial[((int64_t)s * Mpad + m) * 160 + n] = acc[r];
    }
}

// one wave per sample
__global__ void __launch_bounds__(64) mano_pre_bwd_kernel(ManoConst mc, const float* __restrict__ full_pose,
                                                          const float* __restrict__ rots,
                                                          const float* __restrict__ joints, const float* __restrict__ G,
                                                          const float* __restrict__ part,
                                                          const float* __restrict__ partial,
                                                          const float* __restrict__ grad_jtr,
                                                          float* __restrict__ grad_pose, float* __restrict__ grad_betas,
                                                          int B, int Mpad) {
    __shared__ float s_gG2[MN_J * 12], s_gc[3], s_gcoef[MN_KP], s_gG[MN_J * 12], s_gR[MN_J * 9], s_gJ[48], s_gp[48];
    __shared__ float s_G[MN_J * 12], s_R[MN_J * 9], s_J[48];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int np = 3 + mc.ncomps;
    for (int k = lane; k < MN_J * 12; k += 64) {
        float acc = 0.0f;
        for (int c = 0; c < MN_CHUNKS; c++) acc += part[((int64_t)b * MN_CHUNKS + c) * MN_PART + k];
        s_gG2[k] = acc;
        s_G[k] = G[b * MN_J * 12 + k];
    }
    if (lane < 3) {
        float acc = 0.0f;
        for (int c = 0; c < MN_CHUNKS; c++) acc += part[((int64_t)b * MN_CHUNKS + c) * MN_PART + MN_J * 12 + lane];
        s_gc[lane] = acc;
    }
    for (int k = lane; k < MN_KP; k += 64) {
        float acc = 0.0f;
        for (int s = 0; s < MN_SPLITK; s++) acc += partial[((int64_t)s * Mpad + b) * 160 + k];
        s_gcoef[k] = acc;
    }
    for (int k = lane; k < MN_J * 9; k += 64) s_R[k] = rots[b * MN_J * 9 + k];
    if (lane < 48) { s_J[lane] = joints[b * 48 + lane]; s_gJ[lane] = 0.0f; }
    __syncthreads();
    // joint rows of jtr: (G[src].t - centre) * 1000 ; centre gradient gathers everything that was centred
    if (lane == 0) {
        float gt[MN_J][3];
        for (int j = 0; j < MN_J; j++) gt[j][0] = gt[j][1] = gt[j][2] = 0.0f;
        float gc[3] = {s_gc[0], s_gc[1], s_gc[2]};
        if (grad_jtr)
            for (int k = 0; k < MN_JT; k++) {
                const int src = mc.reorder[k];
                for (int r = 0; r < 3; r++) {
                    const float g = grad_jtr[(b * MN_JT + k) * 3 + r] * 1000.0f;
                    if (src < MN_J) { gt[src][r] += g; gc[r] -= g; }
                    // (tips: their share of the centre gradient is in part[] through the vertex rows)
                }
            }
        if (mc.center >= 0)
            for (int r = 0; r < 3; r++) gt[mc.center][r] += gc[r];
        // G2 = [G.R | G.t - G.R J]  ->  dG.R = dG2.R - dG2.t (x) J ; dG.t = dG2.t ; dJ = - G.R^T dG2.t
        for (int j = 0; j < MN_J; j++) {
            for (int r = 0; r < 3; r++) {
                const float gt2 = s_gG2[j * 12 + 4 * r + 3];
                for (int c = 0; c < 3; c++) {
                    s_gG[j * 12 + 4 * r + c] = s_gG2[j * 12 + 4 * r + c] - gt2 * s_J[3 * j + c];
                    s_gJ[3 * j + c] -= s_G[j * 12 + 4 * r + c] * gt2;
                }
                s_gG[j * 12 + 4 * r + 3] = gt2 + gt[j][r];
            }
        }
        // chain, children before parents: G_j = G_p o rel_j, rel_j = [R_j | J_j - J_p]
        for (int j = MN_J - 1; j >= 0; j--) {
            const int pa = mc.parents[j];
            if (pa < 0) {
                for (int r = 0; r < 3; r++) {
                    for (int c = 0; c < 3; c++) s_gR[j * 9 + 3 * r + c] = s_gG[j * 12 + 4 * r + c];
                    s_gJ[3 * j + r] += s_gG[j * 12 + 4 * r + 3];
                }
                continue;
            }
            float rel[12];
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) rel[4 * r + c] = s_R[j * 9 + 3 * r + c];
                rel[4 * r + 3] = s_J[3 * j + r] - s_J[3 * pa + r];
            }
            // d rel = G_p.R^T dG_j ; d G_p.R += dG_j.R rel.R^T + dG_j.t (x) rel.t ; d G_p.t += dG_j.t
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 4; c++) {
                    float acc = 0.0f;
                    for (int k = 0; k < 3; k++) acc += s_G[pa * 12 + 4 * k + r] * s_gG[j * 12 + 4 * k + c];
                    if (c < 3) s_gR[j * 9 + 3 * r + c] = acc;
                    else { s_gJ[3 * j + r] += acc; s_gJ[3 * pa + r] -= acc; }
                }
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) {
                    float acc = 0.0f;
                    for (int k = 0; k < 4; k++) acc += s_gG[j * 12 + 4 * r + k] * rel[4 * c + k];
                    s_gG[pa * 12 + 4 * r + c] += acc;
                }
                s_gG[pa * 12 + 4 * r + 3] += s_gG[j * 12 + 4 * r + 3];
            }
        }
    }
    __syncthreads();
    // pose features: coeff[10 + 9 (j - 1) + k] = R_j[k] - I
    for (int k = lane; k < 135; k += 64) s_gR[9 + k] += s_gcoef[10 + k];
    __syncthreads();
    if (lane < MN_J) {
        float gr[3];
        rodrigues_bwd(&full_pose[b * 48 + 3 * lane], &s_gR[lane * 9], gr);
#pragma unroll
        for (int k = 0; k < 3; k++) s_gp[3 * lane + k] = gr[k];
    }
    __syncthreads();
    if (lane < 3) grad_pose[b * np + lane] = s_gp[lane];
    if (lane < mc.ncomps) {
        float acc = 0.0f;
        for (int l = 0; l < 45; l++) acc = fmaf(mc.comps[lane * 45 + l], s_gp[3 + l], acc);
        grad_pose[b * np + 3 + lane] = acc;
    }
    if (lane < 10) {
        float acc = s_gcoef[lane];
        for (int l = 0; l < 48; l++) acc = fmaf(mc.js[l * 10 + lane], s_gJ[l], acc);
        grad_betas[b * 10 + lane] = acc;
    }
}

}  // namespace mr

using namespace mr;

static ManoConst mano_const(const float* comps, const float* mean, const float* js, const float* jt, const float* blend,
                            const float* templ, const float* weights, const int32_t* parents, const int32_t* tips,
                            const int32_t* reorder, int ncomps, int center) {
    return ManoConst{comps, mean, js, jt, blend, templ, weights, (const int*)parents, (const int*)tips, (const int*)reorder,
                     ncomps, center};
}

extern "C" int64_t mr_mano_workspace_floats(int batch_size) {
    if (batch_size < 0) return MR_ERR_BADARG;
    const int64_t B = batch_size, Mpad = (B + 31) / 32 * 32;
    // coeff | G | G2 | rots | joints | full_pose | v_posed | grad_vp | part | partial
    return B * MN_KP + 2 * B * MN_J * 12 + B * MN_J * 9 + 2 * B * 48 + 2 * B * MN_NV3 + B * MN_CHUNKS * MN_PART +
           (int64_t)MN_SPLITK * Mpad * 160;
}

struct ManoWork {
    float *coeff, *G, *G2, *rots, *joints, *full_pose, *v_posed, *grad_vp, *part, *partial;
    int Mpad;
};
static ManoWork mano_work(float* w, int B) {
    ManoWork m;
    m.Mpad = (B + 31) / 32 * 32;
    m.coeff = w; w += (int64_t)B * MN_KP;
    m.G = w; w += (int64_t)B * MN_J * 12;
    m.G2 = w; w += (int64_t)B * MN_J * 12;
    m.rots = w; w += (int64_t)B * MN_J * 9;
    m.joints = w; w += (int64_t)B * 48;
    m.full_pose = w; w += (int64_t)B * 48;
    m.v_posed = w; w += (int64_t)B * MN_NV3;
    m.grad_vp = w; w += (int64_t)B * MN_NV3;
    m.part = w; w += (int64_t)B * MN_CHUNKS * MN_PART;
    m.partial = w;
    return m;
}

extern "C" int mr_mano_forward(const float* pose_coeffs, const float* betas, const float* comps, const float* hands_mean,
                               const float* js, const float* jt, const float* blend, const float* v_template,
                               const float* weights, const int32_t* parents, const int32_t* tips, const int32_t* reorder,
                               int ncomps, int center, float* workspace, float* verts_out, float* jtr_out,
                               int batch_size, mr_stream_t stream) {
    if (batch_size < 0 || ncomps < 0 || ncomps > 45 || center >= MN_J) return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    if (!pose_coeffs || !betas || !comps || !hands_mean || !js || !jt || !blend || !v_template || !weights || !parents ||
        !tips || !reorder || !workspace || !verts_out || !jtr_out)
        return MR_ERR_BADARG;
    if (batch_size > 65535) return MR_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const ManoConst mc = mano_const(comps, hands_mean, js, jt, blend, v_template, weights, parents, tips, reorder, ncomps,
                                    center);
    const ManoWork m = mano_work(workspace, batch_size);
    hipLaunchKernelGGL(mano_pre_kernel, dim3((unsigned)batch_size), dim3(64), 0, s, mc, pose_coeffs, betas, m.coeff, m.G,
                       m.G2, m.rots, m.joints, m.full_pose, jtr_out, batch_size);
    MR_CHECK_LAUNCH();
    hipLaunchKernelGGL(mano_blend_kernel, dim3((MN_NV3 + 31) / 32, (unsigned)((batch_size + 31) / 32)), dim3(64), 0, s,
                       (const float*)m.coeff, blend, v_template, m.v_posed, batch_size);
    MR_CHECK_LAUNCH();
    hipLaunchKernelGGL(mano_skin_kernel, dim3((MN_V + 255) / 256, (unsigned)batch_size), dim3(256), 0, s, mc,
                       (const float*)m.v_posed, (const float*)m.G, (const float*)m.G2, verts_out, jtr_out, batch_size);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int mr_mano_backward(const float* comps, const float* hands_mean, const float* js, const float* jt,
                                const float* blend, const float* v_template, const float* weights,
                                const int32_t* parents, const int32_t* tips, const int32_t* reorder, int ncomps,
                                int center, float* workspace, const float* grad_verts, const float* grad_jtr,
                                float* grad_pose_coeffs, float* grad_betas, int batch_size, mr_stream_t stream) {
    if (batch_size < 0 || ncomps < 0 || ncomps > 45 || center >= MN_J) return MR_ERR_BADARG;
    if (batch_size == 0) return MR_OK;
    if (!comps || !hands_mean || !js || !jt || !blend || !v_template || !weights || !parents || !tips || !reorder ||
        !workspace || !grad_pose_coeffs || !grad_betas)
        return MR_ERR_BADARG;
    if (batch_size > 65535) return MR_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const ManoConst mc = mano_const(comps, hands_mean, js, jt, blend, v_template, weights, parents, tips, reorder, ncomps,
                                    center);
    const ManoWork m = mano_work(workspace, batch_size);
    hipLaunchKernelGGL(mano_skin_bwd_kernel, dim3(MN_CHUNKS, (unsigned)batch_size), dim3(256), 0, s, mc,
                       (const float*)m.v_posed, (const float*)m.G2, grad_verts, grad_jtr, m.grad_vp, m.part, batch_size);
    MR_CHECK_LAUNCH();
    hipLaunchKernelGGL(mano_blend_bwd_kernel, dim3(5, (unsigned)(m.Mpad / 32), MN_SPLITK), dim3(64), 0, s,
                       (const float*)m.grad_vp, blend, m.partial, batch_size, m.Mpad);
    MR_CHECK_LAUNCH();
    hipLaunchKernelGGL(mano_pre_bwd_kernel, dim3((unsigned)batch_size), dim3(64), 0, s, mc, (const float*)m.full_pose,
                       (const float*)m.rots, (const float*)m.joints, (const float*)m.G, (const float*)m.part,
                       (const float*)m.partial, grad_jtr, grad_pose_coeffs, grad_betas, batch_size, m.Mpad);
    MR_CHECK_LAUNCH();
    return MR_OK;
}
