// frozen_bn.hip -- BatchNorm with frozen statistics fused with the residual add and the ReLU that follow it,
// forward and backward, for gfx950 (the trainer side of SURVEY 8f "f2").
//
// The reference trains with --freeze_batchnorm (trainmeshwarp.py:205-206, 237-240: BatchNorm layers in eval mode,
// affine parameters trainable), so every BN of the ResNet-18 trunk (resnet.py:31-60, 140-175) is the per-channel
// affine map  z = (x - mean) * (weight / sqrt(var + eps)) + bias,  followed by  out = relu(z)  or
// out = relu(z + identity)  (resnet.py:46-58) or nothing (the down-sampling branch).  Stock PyTorch runs that as
// 2 - 3 element-wise kernels forward and a generic batch_norm_backward_kernel + threshold_backward in the backward:
// 10.7 ms of a 42 ms step, streaming 2.5 GB of activations several times at ~2 TB/s.  Here: ONE pass forward
// (read x [, identity], write out) and ONE pass backward (read grad_out, x [, identity]; write grad_x [, grad of the
// identity]; reduce grad_weight / grad_bias), 16-byte accesses.
//
// Work split: a workgroup owns ONE channel and a range of samples, so the channel constants are wave-uniform
// scalars and the two reductions of the backward finish inside the workgroup; per-(channel, range) partial sums
// are combined by a second tiny kernel in a fixed order (deterministic, no float atomics).
// The ReLU mask is recomputed in the backward from x (and the identity) with the forward's exact expression.
#include "mr_common.hpp"

namespace mr {

struct BnParams {
    const float* x;         // [N,C,HW]
    const float* residual;  // [N,C,HW] or NULL
    const float* weight;    // [C]
    const float* bias;
    const float* mean;
    const float* var;
    float eps;
    int relu;
    int N, C, HW, split;    // split = sample ranges per channel
    // forward
    float* y;
    // backward
    const float* grad_y;
    float* grad_x;
    float* grad_residual;   // NULL or [N,C,HW]
    float* partial;         // [2][C][split]: sum g, sum g * (x - mean)
};

__device__ __forceinline__ void channel_consts(const BnParams& p, int c, float& mean, float& a, float& b, float& invstd) {
    mean = p.mean[c];
    invstd = 1.0f / sqrtf(p.var[c] + p.eps);
    a = p.weight[c] * invstd;
    b = p.bias[c];
}

// grid = C * split workgroups of 256 threads; workgroup (c, k) covers samples [k * N / split, (k + 1) * N / split)
template <bool VEC, bool BACKWARD>
__global__ __launch_bounds__(256) void bn_act_kernel(BnParams p) {
    const int c = blockIdx.x / p.split, k = blockIdx.x % p.split;
    const int n0 = (int)((int64_t)k * p.N / p.split), n1 = (int)((int64_t)(k + 1) * p.N / p.split);
    float mean, a, b, invstd;
    channel_consts(p, c, mean, a, b, invstd);
    constexpr int W = VEC ? 4 : 1;
    const int per_plane = p.HW / W;                 // VEC: HW % 4 == 0
    const unsigned total = (unsigned)(n1 - n0) * (unsigned)per_plane;  // < 2^31, checked by the host
    const bool relu = p.relu != 0;
    float sum_g = 0.0f, sum_gx = 0.0f;
    for (unsigned e = threadIdx.x; e < total; e += 256) {
        const unsigned q = e / (unsigned)per_plane;
        const int n = n0 + (int)q, j = (int)(e - q * (unsigned)per_plane);
        const int64_t o = ((int64_t)n * p.C + c) * p.HW + (int64_t)j * W;
        float xv[W], rv[W], gv[W];
        if (VEC) {
            const float4 t = *reinterpret_cast<const float4*>(p.x + o);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            if (p.residual) {
                const float4 r = *reinterpret_cast<const float4*>(p.residual + o);
                rv[0] = r.x; rv[1] = r.y; rv[2] = r.z; rv[3] = r.w;
            }
            if (BACKWARD) {
                const float4 g = *reinterpret_cast<const float4*>(p.grad_y + o);
                gv[0] = g.x; gv[1] = g.y; gv[2] = g.z; gv[3] = g.w;
            }
        } else {
            xv[0] = p.x[o];
            if (p.residual) rv[0] = p.residual[o];
            if (BACKWARD) gv[0] = p.grad_y[o];
        }
        float out[W], gres[W];
#pragma unroll
        for (int i = 0; i < W; i++) {
            const float d = xv[i] - mean;
            float z = d * a + b;
            if (p.residual) z = z + rv[i];
            if (!BACKWARD) {
                out[i] = relu ? fmaxf(z, 0.0f) : z;
            } else {
                const float g = (relu && !(z > 0.0f)) ? 0.0f : gv[i];
                gres[i] = g;
                out[i] = g * a;
                sum_g += g;
                sum_gx += g * d;
            }
        }
        float* dst = BACKWARD ? p.grad_x : p.y;
        if (VEC) {
            *reinterpret_cast<float4*>(dst + o) = make_float4(out[0], out[1], out[2], out[3]);
            if (BACKWARD && p.grad_residual)
                *reinterpret_cast<float4*>(p.grad_residual + o) = make_float4(gres[0], gres[1], gres[2], gres[3]);
        } else {
            dst[o] = out[0];
            if (BACKWARD && p.grad_residual) p.grad_residual[o] = gres[0];
        }
    }
    if (BACKWARD && p.partial) {
        __shared__ float red[2][4];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            sum_g += __shfl_down(sum_g, off);
            sum_gx += __shfl_down(sum_gx, off);
        }
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][wave] = sum_g; red[1][wave] = sum_gx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            p.partial[(int64_t)c * p.split + k] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            p.partial[(int64_t)(p.C + c) * p.split + k] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}

// grad_bias[c] = sum_k partial[0][c][k];  grad_weight[c] = invstd[c] * sum_k partial[1][c][k]   (one wave per channel)
__global__ __launch_bounds__(64) void bn_finish_kernel(const float* __restrict__ partial, const float* __restrict__ var,
                                                       float eps, float* grad_weight, float* grad_bias, int C, int split) {
    const int c = blockIdx.x;
    float s0 = 0.0f, s1 = 0.0f;
    for (int k = threadIdx.x; k < split; k += 64) {
        s0 += partial[(int64_t)c * split + k];
        s1 += partial[(int64_t)(C + c) * split + k];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s0 += __shfl_down(s0, off);
        s1 += __shfl_down(s1, off);
    }
    if (threadIdx.x == 0) {
        if (grad_bias) grad_bias[c] = s0;
        if (grad_weight) grad_weight[c] = s1 * (1.0f / sqrtf(var[c] + eps));
    }
}

// sample ranges per channel: enough workgroups to fill the chip (>= ~4096), never more than N
static inline int bn_split(int N, int C) {
    int s = (4096 + C - 1) / C;
    if (s < 1) s = 1;
    if (s > N) s = N;
    return s;
}

static inline bool bn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace mr

extern "C" int mr_bn_act_forward(const float* x, const float* residual, const float* weight, const float* bias,
                                 const float* running_mean, const float* running_var, float eps, int relu, float* y,
                                 int batch_size, int channels, int plane, mr_stream_t stream) {
    using namespace mr;
    if (batch_size < 0 || channels < 0 || plane < 0) return MR_ERR_BADARG;
    if (batch_size == 0 || channels == 0 || plane == 0) return MR_OK;
    if (!x || !weight || !bias || !running_mean || !running_var || !y) return MR_ERR_BADARG;
    BnParams p{};
    p.x = x; p.residual = residual; p.weight = weight; p.bias = bias; p.mean = running_mean; p.var = running_var;
    p.eps = eps; p.relu = relu; p.N = batch_size; p.C = channels; p.HW = plane; p.split = bn_split(batch_size, channels);
    p.y = y;
    if ((int64_t)channels * p.split > 0x7fffffff || ((int64_t)batch_size / p.split + 1) * plane > 0x7fffffff) return MR_ERR_BADARG;
    const bool vec = plane % 4 == 0 && bn_aligned16(x) && bn_aligned16(y) && bn_aligned16(residual);
    const dim3 grid((unsigned)(channels * p.split));
    if (vec) hipLaunchKernelGGL((bn_act_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((bn_act_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int64_t mr_bn_act_backward_workspace_bytes(int batch_size, int channels) {
    if (batch_size < 0 || channels < 0) return -1;
    return (int64_t)2 * channels * mr::bn_split(batch_size > 0 ? batch_size : 1, channels > 0 ? channels : 1) * 4 + 16;
}

extern "C" int mr_bn_act_backward(const float* grad_y, const float* x, const float* residual, const float* weight,
                                  const float* bias, const float* running_mean, const float* running_var, float eps,
                                  int relu, float* grad_x, float* grad_residual, float* grad_weight, float* grad_bias,
                                  void* workspace, int64_t workspace_bytes, int batch_size, int channels, int plane,
                                  mr_stream_t stream) {
    using namespace mr;
    if (batch_size < 0 || channels < 0 || plane < 0) return MR_ERR_BADARG;
    if (channels == 0) return MR_OK;
    if (!weight || !bias || !running_mean || !running_var) return MR_ERR_BADARG;
    const bool want_params = grad_weight || grad_bias;
    if (batch_size == 0 || plane == 0) {
        hipError_t e = hipSuccess;
        if (grad_weight) e = hipMemsetAsync(grad_weight, 0, (size_t)channels * 4, (hipStream_t)stream);
        if (e == hipSuccess && grad_bias) e = hipMemsetAsync(grad_bias, 0, (size_t)channels * 4, (hipStream_t)stream);
        return e == hipSuccess ? MR_OK : (int)e;
    }
    if (!grad_y || !x || !grad_x) return MR_ERR_BADARG;
    if (grad_residual && !residual) return MR_ERR_BADARG;
    if (want_params && (!workspace || workspace_bytes < mr_bn_act_backward_workspace_bytes(batch_size, channels)))
        return MR_ERR_BADARG;
    BnParams p{};
    p.x = x; p.residual = residual; p.weight = weight; p.bias = bias; p.mean = running_mean; p.var = running_var;
    p.eps = eps; p.relu = relu; p.N = batch_size; p.C = channels; p.HW = plane; p.split = bn_split(batch_size, channels);
    p.grad_y = grad_y; p.grad_x = grad_x; p.grad_residual = grad_residual;
    p.partial = want_params ? static_cast<float*>(workspace) : nullptr;
    if ((int64_t)channels * p.split > 0x7fffffff || ((int64_t)batch_size / p.split + 1) * plane > 0x7fffffff) return MR_ERR_BADARG;
    const bool vec = plane % 4 == 0 && bn_aligned16(x) && bn_aligned16(grad_y) && bn_aligned16(grad_x) &&
                     bn_aligned16(residual) && bn_aligned16(grad_residual);
    const dim3 grid((unsigned)(channels * p.split));
    if (vec) hipLaunchKernelGGL((bn_act_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((bn_act_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    if (want_params) {
        hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)channels), dim3(64), 0, (hipStream_t)stream, p.partial,
                           running_var, eps, grad_weight, grad_bias, channels, p.split);
        MR_CHECK_LAUNCH();
    }
    return MR_OK;
}
