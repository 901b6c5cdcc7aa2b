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

// ReLU that propagates NaN like torch.relu (fmaxf(NaN, 0) would return 0 and hide a diverged trunk from the
// "Loss became nan!" guard)
__device__ __forceinline__ float relu_nan(float z) { return z > 0.0f ? z : (z != z ? z : 0.0f); }


struct BnParams {
    const void* x;          // [N,C,HW]  fp32 or bf16 (the activation type T of the kernel)
    const void* residual;   // [N,C,HW] or NULL
    const float* weight;    // [C]
    const float* bias;
    const float* mean;
    const float* var;
    float eps;
    int relu;
    int N, C, HW, split;    // split = sample ranges per channel
    // forward
    void* y;
    // backward
    const void* grad_y;
    const void* grad_y2;    // optional second gradient of y (y feeds two consumers): summed on load
    void* grad_x;
    void* grad_residual;    // NULL or [N,C,HW]
    float* partial;         // [2][C][split]: sum g, sum g * (x - mean)
};

// activations: fp32, or bf16 (the trunk under bf16 autocast) converted on load / rounded to nearest-even on store
typedef unsigned short bf16_t;
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<bf16_t> { typedef ushort4 type; };

template <typename T>
__device__ __forceinline__ void load4(const void* base, int64_t o, float* v) {
    const typename Vec4<T>::type t = *reinterpret_cast<const typename Vec4<T>::type*>(static_cast<const T*>(base) + o);
    v[0] = to_f32(t.x); v[1] = to_f32(t.y); v[2] = to_f32(t.z); v[3] = to_f32(t.w);
}
template <typename T>
__device__ __forceinline__ void store4(void* base, int64_t o, const float* v) {
    typename Vec4<T>::type t;
    t.x = from_f32<T>(v[0]); t.y = from_f32<T>(v[1]); t.z = from_f32<T>(v[2]); t.w = from_f32<T>(v[3]);
    *reinterpret_cast<typename Vec4<T>::type*>(static_cast<T*>(base) + o) = t;
}

__device__ __forceinline__ void channel_consts(const BnParams& p, int c, float& mean, float& a, float& b, float& invstd) {
    mean = p.mean[c];
    invstd = 1.0f / sqrtf(p.var[c] + p.eps);
    a = p.weight[c] * invstd;
    b = p.bias[c];
}

// grid = C * split workgroups of 256 threads; workgroup (c, k) covers samples [k * N / split, (k + 1) * N / split)
template <typename T, bool VEC, bool BACKWARD>
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
            load4<T>(p.x, o, xv);
            if (p.residual) load4<T>(p.residual, o, rv);
            if (BACKWARD) {
                load4<T>(p.grad_y, o, gv);
                if (p.grad_y2) {
                    float g2[4];
                    load4<T>(p.grad_y2, o, g2);
#pragma unroll
                    for (int i = 0; i < 4; i++) gv[i] += g2[i];
                }
            }
        } else {
            xv[0] = to_f32(static_cast<const T*>(p.x)[o]);
            if (p.residual) rv[0] = to_f32(static_cast<const T*>(p.residual)[o]);
            if (BACKWARD) {
                gv[0] = to_f32(static_cast<const T*>(p.grad_y)[o]);
                if (p.grad_y2) gv[0] += to_f32(static_cast<const T*>(p.grad_y2)[o]);
            }
        }
        float out[W], gres[W];
#pragma unroll
        for (int i = 0; i < W; i++) {
            const float d = xv[i] - mean;
            float z = d * a + b;
            if (p.residual) z = z + rv[i];
            if (!BACKWARD) {
                out[i] = relu ? relu_nan(z) : z;
            } else {
                const float g = (relu && !(z > 0.0f)) ? 0.0f : gv[i];
                gres[i] = g;
                out[i] = g * a;
                sum_g += g;
                sum_gx += g * d;
            }
        }
        void* dst = BACKWARD ? p.grad_x : p.y;
        if (VEC) {
            store4<T>(dst, o, out);
            if (BACKWARD && p.grad_residual) store4<T>(p.grad_residual, o, gres);
        } else {
            static_cast<T*>(dst)[o] = from_f32<T>(out[0]);
            if (BACKWARD && p.grad_residual) static_cast<T*>(p.grad_residual)[o] = from_f32<T>(gres[0]);
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

// grad_bias[c] = sum_k partial[0][c][k];  grad_weight[c] = invstd[c] * sum_k partial[1][c][k]   (one workgroup per channel,
// fixed summation order)
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ partial, const float* __restrict__ var,
                                                        float eps, float* grad_weight, float* grad_bias, int C, int split) {
    __shared__ float red[2][4];
    const int c = blockIdx.x;
    float s0 = 0.0f, s1 = 0.0f;
    for (int k = threadIdx.x; k < split; k += 256) {
        s0 += partial[(int64_t)c * split + k];
        s1 += partial[(int64_t)(C + c) * split + k];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s0 += __shfl_down(s0, off);
        s1 += __shfl_down(s1, off);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (grad_bias) grad_bias[c] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        if (grad_weight) grad_weight[c] = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * (1.0f / sqrtf(var[c] + eps));
    }
}

// ---- channels-last (NHWC) activations: memory order [N, H*W, C] --------------------------------------------
// MIOpen's convolutions are faster on channels-last tensors (no NCHW<->NHWC transposes around its implicit-GEMM
// kernels: 23.5 instead of 26.9 ms per step for the trunk's convolutions, scripts/conv_layout.py), so the glue
// kernels come in that layout too.  A thread owns FOUR consecutive channels (one 16-byte access) and walks pixels:
// with 1024 % C == 0 its channel group never changes, so the channel constants live in registers and the backward's
// reductions are per-thread partial sums combined once per workgroup; workgroup partials [blocks][C] are summed by
// bn_finish_kernel in a fixed order.
constexpr int BN_NHWC_BLOCKS = 2048;

template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) void bn_act_nhwc_kernel(BnParams p) {
    __shared__ float red[256][9];
    const int groups = p.C >> 2;                       // float4 groups per pixel; 256 % groups == 0
    const int cg = threadIdx.x % groups, prow = threadIdx.x / groups, rows = 256 / groups;
    const int c0 = 4 * cg;
    float mean[4], a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float invstd;
        channel_consts(p, c0 + i, mean[i], a[i], b[i], invstd);
    }
    const int64_t pixels = (int64_t)p.N * p.HW;
    const bool relu = p.relu != 0;
    float sg[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sgx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t px = (int64_t)blockIdx.x * rows + prow; px < pixels; px += (int64_t)gridDim.x * rows) {
        const int64_t o = px * p.C + c0;
        float xv[4], rv[4], gv[4], out[4], gres[4];
        load4<T>(p.x, o, xv);
        if (p.residual) load4<T>(p.residual, o, rv);
        if (BACKWARD) {
            load4<T>(p.grad_y, o, gv);
            if (p.grad_y2) {
                float g2[4];
                load4<T>(p.grad_y2, o, g2);
#pragma unroll
                for (int i = 0; i < 4; i++) gv[i] += g2[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float d = xv[i] - mean[i];
            float z = d * a[i] + b[i];
            if (p.residual) z = z + rv[i];
            if (!BACKWARD) {
                out[i] = relu ? relu_nan(z) : z;
            } else {
                const float g = (relu && !(z > 0.0f)) ? 0.0f : gv[i];
                gres[i] = g;
                out[i] = g * a[i];
                sg[i] += g;
                sgx[i] += g * d;
            }
        }
        store4<T>(BACKWARD ? p.grad_x : p.y, o, out);
        if (BACKWARD && p.grad_residual) store4<T>(p.grad_residual, o, gres);
    }
    if (BACKWARD && p.partial) {
#pragma unroll
        for (int i = 0; i < 4; i++) { red[threadIdx.x][i] = sg[i]; red[threadIdx.x][4 + i] = sgx[i]; }
        __syncthreads();
        if (threadIdx.x < groups) {
            float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < rows; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) t[i] += red[r * groups + threadIdx.x][i];
            // partial layout [2][C][blocks] (the finish kernel's), slot = this workgroup
#pragma unroll
            for (int i = 0; i < 4; i++) {
                p.partial[(int64_t)(c0 + i) * gridDim.x + blockIdx.x] = t[i];
                p.partial[(int64_t)(p.C + c0 + i) * gridDim.x + blockIdx.x] = t[4 + i];
            }
        }
    }
}

static inline bool bn_nhwc_ok(int C) { return C >= 4 && C <= 1024 && (1024 % C) == 0; }

static inline int bn_nhwc_blocks(int64_t pixels, int C) {
    const int rows = 256 / (C / 4);
    const int64_t need = (pixels + rows - 1) / rows;
    return (int)(need < BN_NHWC_BLOCKS ? (need < 1 ? 1 : need) : BN_NHWC_BLOCKS);
}

// sample ranges per channel: enough workgroups to fill the chip (>= ~4096), never more than N
static inline int bn_split(int N, int C) {
    int s = (4096 + C - 1) / C;
    if (s < 1) s = 1;
    if (s > N) s = N;
    return s;
}

static inline bool bn_aligned(const void* p, int bytes) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(bytes - 1)) == 0; }

template <bool BACKWARD>
static void bn_launch(const BnParams& p, int act_dtype, bool vec, hipStream_t s) {
    const dim3 grid((unsigned)(p.C * p.split));
    if (act_dtype == 0) {
        if (vec) hipLaunchKernelGGL((bn_act_kernel<float, true, BACKWARD>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((bn_act_kernel<float, false, BACKWARD>), grid, dim3(256), 0, s, p);
    } else {
        if (vec) hipLaunchKernelGGL((bn_act_kernel<bf16_t, true, BACKWARD>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((bn_act_kernel<bf16_t, false, BACKWARD>), grid, dim3(256), 0, s, p);
    }
}

}  // namespace mr

extern "C" int mr_bn_act_forward(const void* x, const void* residual, const float* weight, const float* bias,
                                 const float* running_mean, const float* running_var, float eps, int relu, int act_dtype,
                                 int channels_last, void* y, int batch_size, int channels, int plane,
                                 mr_stream_t stream) {
    using namespace mr;
    if (batch_size < 0 || channels < 0 || plane < 0 || (act_dtype != 0 && act_dtype != 1)) return MR_ERR_BADARG;
    if (channels_last && channels > 0 && !bn_nhwc_ok(channels)) return MR_ERR_BADARG;
    if (batch_size == 0 || channels == 0 || plane == 0) return MR_OK;
    if (!x || !weight || !bias || !running_mean || !running_var || !y) return MR_ERR_BADARG;
    BnParams p{};
    p.x = x; p.residual = residual; p.weight = weight; p.bias = bias; p.mean = running_mean; p.var = running_var;
    p.eps = eps; p.relu = relu; p.N = batch_size; p.C = channels; p.HW = plane; p.split = bn_split(batch_size, channels);
    p.y = y;
    if ((int64_t)channels * p.split > 0x7fffffff || ((int64_t)batch_size / p.split + 1) * plane > 0x7fffffff) return MR_ERR_BADARG;
    const int vb = act_dtype == 0 ? 16 : 8;  // bytes of a 4-element access
    if (channels_last) {
        if (!bn_aligned(x, vb) || !bn_aligned(y, vb) || !bn_aligned(residual, vb)) return MR_ERR_BADARG;
        const dim3 grid((unsigned)bn_nhwc_blocks((int64_t)batch_size * plane, channels));
        if (act_dtype == 0) hipLaunchKernelGGL((bn_act_nhwc_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((bn_act_nhwc_kernel<bf16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        MR_CHECK_LAUNCH();
        return MR_OK;
    }
    const bool vec = plane % 4 == 0 && bn_aligned(x, vb) && bn_aligned(y, vb) && bn_aligned(residual, vb);
    bn_launch<false>(p, act_dtype, vec, (hipStream_t)stream);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int64_t mr_bn_act_backward_workspace_bytes(int batch_size, int channels) {
    if (batch_size < 0 || channels < 0) return -1;
    const int64_t split = mr::bn_split(batch_size > 0 ? batch_size : 1, channels > 0 ? channels : 1);
    return (int64_t)2 * channels * (split > mr::BN_NHWC_BLOCKS ? split : mr::BN_NHWC_BLOCKS) * 4 + 16;
}

extern "C" int mr_bn_act_backward(const void* grad_y, const void* grad_y2, const void* x, const void* residual,
                                  const float* weight,
                                  const float* bias, const float* running_mean, const float* running_var, float eps,
                                  int relu, int act_dtype, int channels_last, void* grad_x, void* grad_residual,
                                  float* grad_weight, float* grad_bias, void* workspace, int64_t workspace_bytes,
                                  int batch_size, int channels, int plane, mr_stream_t stream) {
    using namespace mr;
    if (batch_size < 0 || channels < 0 || plane < 0 || (act_dtype != 0 && act_dtype != 1)) return MR_ERR_BADARG;
    if (channels_last && channels > 0 && !bn_nhwc_ok(channels)) return MR_ERR_BADARG;
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
    p.grad_y = grad_y; p.grad_y2 = grad_y2; p.grad_x = grad_x; p.grad_residual = grad_residual;
    p.partial = want_params ? static_cast<float*>(workspace) : nullptr;
    if ((int64_t)channels * p.split > 0x7fffffff || ((int64_t)batch_size / p.split + 1) * plane > 0x7fffffff) return MR_ERR_BADARG;
    const int vb = act_dtype == 0 ? 16 : 8;
    int slots = p.split;  // partial sums per channel
    if (channels_last) {
        if (!bn_aligned(x, vb) || !bn_aligned(grad_y, vb) || !bn_aligned(grad_y2, vb) || !bn_aligned(grad_x, vb) ||
            !bn_aligned(residual, vb) ||
            !bn_aligned(grad_residual, vb))
            return MR_ERR_BADARG;
        slots = bn_nhwc_blocks((int64_t)batch_size * plane, channels);
        const dim3 grid((unsigned)slots);
        if (act_dtype == 0) hipLaunchKernelGGL((bn_act_nhwc_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((bn_act_nhwc_kernel<bf16_t, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        const bool vec = plane % 4 == 0 && bn_aligned(x, vb) && bn_aligned(grad_y, vb) && bn_aligned(grad_y2, vb) &&
                         bn_aligned(grad_x, vb) &&
                         bn_aligned(residual, vb) && bn_aligned(grad_residual, vb);
        bn_launch<true>(p, act_dtype, vec, (hipStream_t)stream);
    }
    MR_CHECK_LAUNCH();
    if (want_params) {
        hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)channels), dim3(256), 0, (hipStream_t)stream, p.partial,
                           running_var, eps, grad_weight, grad_bias, channels, slots);
        MR_CHECK_LAUNCH();
    }
    return MR_OK;
}
