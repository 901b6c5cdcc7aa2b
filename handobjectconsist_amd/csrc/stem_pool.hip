// stem_pool.hip -- the ResNet stem after its 7x7 convolution, fused: BatchNorm with frozen statistics -> ReLU ->
// MaxPool2d(kernel 3, stride 2, padding 1) (resnet.py:140-147 with --freeze_batchnorm), forward and backward.
//
// Stock PyTorch writes relu(bn(x)) (805 MB at B = 3 x 64, 256 x 256 inputs), reads it back to pool it, keeps it and a
// 64-bit index map for the backward, and runs max_pool_backward (1.46 ms) + batch_norm_backward on top: 3.0 ms of a
// 34 ms step.  Here the full-resolution activation after the BN never exists: the forward reads the convolution
// output once and writes the pooled map (1/4 of the pixels); the backward recomputes z = bn(x) per tile in LDS,
// re-derives every window's arg-max (PyTorch's rule: scan kh, kw ascending, strictly-greater wins, padding ignored)
// and GATHERS the pooled gradient per input pixel -- no atomics, no zero-fill of the 805 MB gradient.
#include "mr_common.hpp"

namespace mr {

// ReLU / max that propagate NaN like torch.relu / max_pool2d ("val > max || isnan(val)")
__device__ __forceinline__ float sp_relu_nan(float z) { return z > 0.0f ? z : (z != z ? z : 0.0f); }


struct StemParams {
    const void* x;        // [N,C,H,W] convolution output, fp32 or bf16 (the activation type T of the kernels)
    const float* weight;  // [C]
    const float* bias;
    const float* mean;
    const float* var;
    float eps;
    int N, C, H, W, OH, OW;   // OH = (H - 1) / 2 + 1
    int tiles_x, tiles_y;
    void* y;              // forward: [N,C,OH,OW]
    const void* grad_y;   // backward
    const void* grad_y2;  // optional second gradient of y (y feeds two consumers): summed on load
    void* grad_x;         // [N,C,H,W]
    float* partial;       // [2][C][N * tiles]: sum g, sum g * (x - mean)   (channels-last: [2][C][workgroups])
    unsigned char* argmax;  // channels-last only: [N,OH,OW,C] arg-max position kh * 3 + kw, written by the forward
};

constexpr int SP_TX = 32, SP_TY = 16;  // windows (= pooled pixels) per tile
// Input region of a tile in LDS: rows 2 oy0 - 1 .. 2 (oy0 + TY) + 1, columns 2 ox0 - 4 .. 2 (ox0 + TX) + 3 -- the
// column range starts 3 pixels early so that it begins on a 16-byte boundary (2 ox0 is a multiple of 64) and can be
// fetched with aligned float4 loads.  LDS row = input row - (2 oy0 - 1), LDS column = input column - (2 ox0 - 4).
constexpr int SP_IH = 2 * SP_TY + 3, SP_IW = 2 * SP_TX + 8, SP_C0 = 4;
constexpr int SP_LDW = SP_IW + 1;

typedef unsigned short bf16_t;
__device__ __forceinline__ float sp_f32(float v) { return v; }
__device__ __forceinline__ float sp_f32(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
template <typename T> __device__ __forceinline__ T sp_from(float v);
template <> __device__ __forceinline__ float sp_from<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t sp_from<bf16_t>(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
    return (bf16_t)(u >> 16);
}
template <typename T> struct SpVec4;
template <> struct SpVec4<float> { typedef float4 type; };
template <> struct SpVec4<bf16_t> { typedef ushort4 type; };

__device__ __forceinline__ void stem_consts(const StemParams& p, int c, float& mean, float& a, float& b) {
    mean = p.mean[c];
    a = p.weight[c] * (1.0f / sqrtf(p.var[c] + p.eps));
    b = p.bias[c];
}

// z = bn(x) (and d = x - mean) of the tile's input region into LDS; -inf / 0 outside the image
template <typename T, bool KEEP_D>
__device__ __forceinline__ void load_tile(const StemParams& p, const T* xp, int oy0, int ox0, float mean, float a,
                                          float b, float (*zt)[SP_LDW], float (*dt)[SP_LDW]) {
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - SP_C0;
    const float ninf = -__builtin_inff();
    if ((p.W & 3) == 0 && (reinterpret_cast<uintptr_t>(xp) & (4 * sizeof(T) - 1)) == 0) {
        for (int e = threadIdx.x; e < SP_IH * (SP_IW / 4); e += 256) {
            const int r = e / (SP_IW / 4), q = e - r * (SP_IW / 4);
            const int iy = iy0 + r, ix = ix0 + 4 * q;  // a float4 is entirely inside or entirely outside the row
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            if (in) {
                const typename SpVec4<T>::type t = *reinterpret_cast<const typename SpVec4<T>::type*>(xp + (int64_t)iy * p.W + ix);
                v[0] = sp_f32(t.x); v[1] = sp_f32(t.y); v[2] = sp_f32(t.z); v[3] = sp_f32(t.w);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float d = v[i] - mean;
                zt[r][4 * q + i] = in ? d * a + b : ninf;
                if (KEEP_D) dt[r][4 * q + i] = in ? d : 0.0f;
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < SP_IH * SP_IW; e += 256) {
        const int r = e / SP_IW, c = e - r * SP_IW;
        const int iy = iy0 + r, ix = ix0 + c;
        float z = ninf, d = 0.0f;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
            d = sp_f32(xp[(int64_t)iy * p.W + ix]) - mean;
            z = d * a + b;
        }
        zt[r][c] = z;
        if (KEEP_D) dt[r][c] = d;
    }
}

// grid = N * C * tiles workgroups of 256 threads
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_forward_kernel(StemParams p) {
    __shared__ float zt[SP_IH][SP_LDW];
    const int tiles = p.tiles_x * p.tiles_y;
    const int plane = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int c = plane % p.C;
    const int oy0 = (t / p.tiles_x) * SP_TY, ox0 = (t % p.tiles_x) * SP_TX;
    float mean, a, b;
    stem_consts(p, c, mean, a, b);
    load_tile<T, false>(p, static_cast<const T*>(p.x) + (int64_t)plane * p.H * p.W, oy0, ox0, mean, a, b, zt, nullptr);
    __syncthreads();
    for (int e = threadIdx.x; e < SP_TY * SP_TX; e += 256) {
        const int wy = e / SP_TX, wx = e % SP_TX;
        const int oy = oy0 + wy, ox = ox0 + wx;
        if (oy >= p.OH || ox >= p.OW) continue;
        float m = 0.0f;  // relu: max(0, max z); every window holds at least one pixel of the image
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const float z = zt[2 * wy + kh][2 * wx + kw + SP_C0 - 1];
                m = (z > m || z != z) ? z : m;
            }
        static_cast<T*>(p.y)[((int64_t)plane * p.OH + oy) * p.OW + ox] = sp_from<T>(m);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_pool_backward_kernel(StemParams p) {
    __shared__ float zt[SP_IH][SP_LDW];
    __shared__ float dt[SP_IH][SP_LDW];                 // x - mean of the same region (for grad_weight)
    __shared__ float gw[SP_TY + 1][SP_TX + 1];          // pooled gradient of the tile's windows (+1 row / column)
    __shared__ unsigned char am[SP_TY + 1][SP_TX + 1];  // arg-max position kh * 3 + kw of every window
    __shared__ float red[2][4];
    const int tiles = p.tiles_x * p.tiles_y;
    const int plane = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int c = plane % p.C;
    const int oy0 = (t / p.tiles_x) * SP_TY, ox0 = (t % p.tiles_x) * SP_TX;
    float mean, a, b;
    stem_consts(p, c, mean, a, b);
    const T* xp = static_cast<const T*>(p.x) + (int64_t)plane * p.H * p.W;
    load_tile<T, true>(p, xp, oy0, ox0, mean, a, b, zt, dt);
    __syncthreads();
    // windows oy0 .. oy0 + TY, ox0 .. ox0 + TX: the owned input rows 2 oy0 .. 2 (oy0 + TY) - 1 touch one window more
    for (int e = threadIdx.x; e < (SP_TY + 1) * (SP_TX + 1); e += 256) {
        const int wy = e / (SP_TX + 1), wx = e - wy * (SP_TX + 1);
        const int oy = oy0 + wy, ox = ox0 + wx;
        float g = 0.0f;
        int best = 0;
        if (oy < p.OH && ox < p.OW) {
            g = sp_f32(static_cast<const T*>(p.grad_y)[((int64_t)plane * p.OH + oy) * p.OW + ox]);
            if (p.grad_y2) g += sp_f32(static_cast<const T*>(p.grad_y2)[((int64_t)plane * p.OH + oy) * p.OW + ox]);
            float mv = -__builtin_inff();  // PyTorch: first strictly greater value of relu(z) in (kh, kw) order, padding skipped
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const float z = zt[2 * wy + k / 3][2 * wx + k % 3 + SP_C0 - 1];
                const float v = sp_relu_nan(z);
                if (z != -__builtin_inff() && (v > mv || v != v)) { mv = v; best = k; }
            }
        }
        gw[wy][wx] = g;
        am[wy][wx] = (unsigned char)best;
    }
    __syncthreads();
    // gather: owned input pixels rows 2 oy0 + [0, 2 TY), columns 2 ox0 + [0, 2 TX); LDS row = input row - (2 oy0 - 1)
    float sum_g = 0.0f, sum_gx = 0.0f;
    T* gx = static_cast<T*>(p.grad_x) + (int64_t)plane * p.H * p.W;
    const bool vec = (p.W & 3) == 0 && (reinterpret_cast<uintptr_t>(gx) & (4 * sizeof(T) - 1)) == 0;
    for (int e = threadIdx.x; e < 2 * SP_TY * (2 * SP_TX / 4); e += 256) {  // four consecutive owned pixels per item
        const int ry = e / (2 * SP_TX / 4), rx0 = 4 * (e - ry * (2 * SP_TX / 4));
        const int iy = 2 * oy0 + ry, ix0 = 2 * ox0 + rx0;
        if (iy >= p.H || ix0 >= p.W) continue;
        const int ly = ry + 1;  // position relative to the first window's first row
        float out[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int lx = rx0 + i + 1;
            // windows containing the pixel: wy with 2 wy <= ly <= 2 wy + 2, likewise wx
            float g = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 2; dy++) {
                const int wy = (ly >> 1) - dy;
                const int kh = ly - 2 * wy;
                if (wy < 0 || kh > 2) continue;
#pragma unroll
                for (int dx = 0; dx < 2; dx++) {
                    const int wx = (lx >> 1) - dx;
                    const int kw = lx - 2 * wx;
                    if (wx < 0 || kw > 2) continue;
                    if (am[wy][wx] == kh * 3 + kw) g += gw[wy][wx];
                }
            }
            const float z = zt[ly][lx + SP_C0 - 1];
            const float gm = (z > 0.0f && ix0 + i < p.W) ? g : 0.0f;  // ReLU
            out[i] = gm * a;
            sum_g += gm;
            sum_gx += gm * dt[ly][lx + SP_C0 - 1];
        }
        T* dst = gx + (int64_t)iy * p.W + ix0;
        if (vec) {
            typename SpVec4<T>::type t;
            t.x = sp_from<T>(out[0]); t.y = sp_from<T>(out[1]); t.z = sp_from<T>(out[2]); t.w = sp_from<T>(out[3]);
            *reinterpret_cast<typename SpVec4<T>::type*>(dst) = t;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (ix0 + i < p.W) dst[i] = sp_from<T>(out[i]);
        }
    }
    if (p.partial) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            sum_g += __shfl_down(sum_g, off);
            sum_gx += __shfl_down(sum_gx, off);
        }
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][wave] = sum_g; red[1][wave] = sum_gx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n = plane / p.C;
            const int64_t per_c = (int64_t)p.N * tiles;
            const int64_t slot = (int64_t)n * tiles + t;
            p.partial[(int64_t)c * per_c + slot] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            p.partial[((int64_t)p.C + c) * per_c + slot] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}

// grad_bias[c] = sum partial[0][c][:];  grad_weight[c] = invstd[c] * sum partial[1][c][:]   (one workgroup per channel)
__global__ __launch_bounds__(256) void stem_finish_kernel(const float* __restrict__ partial, const float* __restrict__ var,
                                                          float eps, float* grad_weight, float* grad_bias, int C,
                                                          int64_t per_c) {
    __shared__ float red[2][4];
    const int c = blockIdx.x;
    float s0 = 0.0f, s1 = 0.0f;
    for (int64_t k = threadIdx.x; k < per_c; k += 256) {
        s0 += partial[(int64_t)c * per_c + k];
        s1 += partial[((int64_t)C + c) * per_c + k];
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

// ---- channels-last (NHWC) activations ---------------------------------------------------------------------
// A thread owns four consecutive channels of one pixel (one 16-byte access; the C / 4 threads of a pixel read a
// contiguous 4 C-byte segment), so no LDS staging is needed: the forward reads the <= 9 pixels of its window directly
// and also writes each channel's arg-max position (1 byte per pooled value); the backward walks INPUT pixels and
// gathers from the <= 4 windows that contain them by comparing the stored positions -- same arithmetic and tie rule
// as the NCHW kernels.
constexpr int SP_NHWC_BLOCKS = 4096;

template <typename T>
__device__ __forceinline__ void sp_load4(const T* base, int64_t o, float* v) {
    const typename SpVec4<T>::type t = *reinterpret_cast<const typename SpVec4<T>::type*>(base + o);
    v[0] = sp_f32(t.x); v[1] = sp_f32(t.y); v[2] = sp_f32(t.z); v[3] = sp_f32(t.w);
}
template <typename T>
__device__ __forceinline__ void sp_store4(T* base, int64_t o, const float* v) {
    typename SpVec4<T>::type t;
    t.x = sp_from<T>(v[0]); t.y = sp_from<T>(v[1]); t.z = sp_from<T>(v[2]); t.w = sp_from<T>(v[3]);
    *reinterpret_cast<typename SpVec4<T>::type*>(base + o) = t;
}

template <typename T>
__global__ __launch_bounds__(256) void stem_pool_nhwc_forward_kernel(StemParams p) {
    const int groups = p.C >> 2, cg = threadIdx.x % groups, prow = threadIdx.x / groups, rows = 256 / groups;
    const int c0 = 4 * cg;
    float mean[4], a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) stem_consts(p, c0 + i, mean[i], a[i], b[i]);
    const T* x = static_cast<const T*>(p.x);
    const int64_t total = (int64_t)p.N * p.OH * p.OW;
    for (int64_t op = (int64_t)blockIdx.x * rows + prow; op < total; op += (int64_t)gridDim.x * rows) {
        const int ox = (int)(op % p.OW), oy = (int)((op / p.OW) % p.OH), n = (int)(op / ((int64_t)p.OW * p.OH));
        float best[4];
        unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; i++) best[i] = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
            if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) continue;
            float xv[4];
            sp_load4<T>(x, (((int64_t)n * p.H + iy) * p.W + ix) * p.C + c0, xv);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float v = sp_relu_nan((xv[i] - mean[i]) * a[i] + b[i]);
                if (v > best[i] || v != v) { best[i] = v; bi[i] = (unsigned)k; }
            }
        }
        const int64_t o = op * p.C + c0;
        sp_store4<T>(static_cast<T*>(p.y), o, best);
        *reinterpret_cast<uchar4*>(p.argmax + o) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1],
                                                               (unsigned char)bi[2], (unsigned char)bi[3]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_pool_nhwc_backward_kernel(StemParams p) {
    __shared__ float red[256][9];
    const int groups = p.C >> 2, cg = threadIdx.x % groups, prow = threadIdx.x / groups, rows = 256 / groups;
    const int c0 = 4 * cg;
    float mean[4], a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) stem_consts(p, c0 + i, mean[i], a[i], b[i]);
    const T* x = static_cast<const T*>(p.x);
    const T* gy = static_cast<const T*>(p.grad_y);
    const T* gy2 = static_cast<const T*>(p.grad_y2);
    const int64_t total = (int64_t)p.N * p.H * p.W;
    float sg[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sgx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t ip = (int64_t)blockIdx.x * rows + prow; ip < total; ip += (int64_t)gridDim.x * rows) {
        const int ix = (int)(ip % p.W), iy = (int)((ip / p.W) % p.H), n = (int)(ip / ((int64_t)p.W * p.H));
        float xv[4], g[4] = {0.0f, 0.0f, 0.0f, 0.0f}, out[4];
        sp_load4<T>(x, ip * p.C + c0, xv);
        const int ly = iy + 1, lx = ix + 1;  // window w covers l = 2 w .. 2 w + 2
#pragma unroll
        for (int dy = 0; dy < 2; dy++) {
            const int wy = (ly >> 1) - dy, kh = ly - 2 * wy;
            if (wy < 0 || wy >= p.OH || kh > 2) continue;
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                const int wx = (lx >> 1) - dx, kw = lx - 2 * wx;
                if (wx < 0 || wx >= p.OW || kw > 2) continue;
                const int64_t o = (((int64_t)n * p.OH + wy) * p.OW + wx) * p.C + c0;
                const uchar4 id = *reinterpret_cast<const uchar4*>(p.argmax + o);
                float gv[4];
                sp_load4<T>(gy, o, gv);
                if (gy2) {
                    float g2[4];
                    sp_load4<T>(gy2, o, g2);
#pragma unroll
                    for (int i = 0; i < 4; i++) gv[i] += g2[i];
                }
                const unsigned code = (unsigned)(kh * 3 + kw);
                g[0] += id.x == code ? gv[0] : 0.0f;
                g[1] += id.y == code ? gv[1] : 0.0f;
                g[2] += id.z == code ? gv[2] : 0.0f;
                g[3] += id.w == code ? gv[3] : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float d = xv[i] - mean[i];
            const float gm = (d * a[i] + b[i] > 0.0f) ? g[i] : 0.0f;  // ReLU
            out[i] = gm * a[i];
            sg[i] += gm;
            sgx[i] += gm * d;
        }
        sp_store4<T>(static_cast<T*>(p.grad_x), ip * p.C + c0, out);
    }
    if (p.partial) {
#pragma unroll
        for (int i = 0; i < 4; i++) { red[threadIdx.x][i] = sg[i]; red[threadIdx.x][4 + i] = sgx[i]; }
        __syncthreads();
        if (threadIdx.x < groups) {
            float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < rows; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) t[i] += red[r * groups + threadIdx.x][i];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                p.partial[(int64_t)(c0 + i) * gridDim.x + blockIdx.x] = t[i];
                p.partial[(int64_t)(p.C + c0 + i) * gridDim.x + blockIdx.x] = t[4 + i];
            }
        }
    }
}

static inline bool sp_nhwc_ok(int C) { return C >= 4 && C <= 1024 && (1024 % C) == 0; }
static inline int sp_nhwc_blocks(int64_t pixels, int C) {
    const int rows = 256 / (C / 4);
    const int64_t need = (pixels + rows - 1) / rows;
    return (int)(need < SP_NHWC_BLOCKS ? (need < 1 ? 1 : need) : SP_NHWC_BLOCKS);
}

static int stem_fill(StemParams& p, const void* x, const float* weight, const float* bias, const float* mean,
                     const float* var, float eps, int N, int C, int H, int W) {
    if (N < 0 || C < 0 || H < 0 || W < 0) return MR_ERR_BADARG;
    p.x = x; p.weight = weight; p.bias = bias; p.mean = mean; p.var = var; p.eps = eps;
    p.N = N; p.C = C; p.H = H; p.W = W;
    p.OH = H > 0 ? (H - 1) / 2 + 1 : 0;
    p.OW = W > 0 ? (W - 1) / 2 + 1 : 0;
    p.tiles_x = (p.OW + SP_TX - 1) / SP_TX;
    p.tiles_y = (p.OH + SP_TY - 1) / SP_TY;
    if ((int64_t)N * C * p.tiles_x * p.tiles_y > 0x7fffffff) return MR_ERR_BADARG;
    return MR_OK;
}

}  // namespace mr

extern "C" int mr_stem_pool_forward(const void* x, const float* weight, const float* bias, const float* running_mean,
                                    const float* running_var, float eps, int act_dtype, int channels_last, void* y,
                                    unsigned char* argmax, int batch_size, int channels, int height, int width,
                                    mr_stream_t stream) {
    using namespace mr;
    if (act_dtype != 0 && act_dtype != 1) return MR_ERR_BADARG;
    if (channels_last && channels > 0 && !sp_nhwc_ok(channels)) return MR_ERR_BADARG;
    StemParams p{};
    const int rc = stem_fill(p, x, weight, bias, running_mean, running_var, eps, batch_size, channels, height, width);
    if (rc != MR_OK) return rc;
    if (batch_size == 0 || channels == 0 || height == 0 || width == 0) return MR_OK;
    if (!x || !weight || !bias || !running_mean || !running_var || !y) return MR_ERR_BADARG;
    p.y = y;
    if (channels_last) {
        const uintptr_t am = (uintptr_t)(act_dtype == 0 ? 15 : 7);
        if (!argmax || (reinterpret_cast<uintptr_t>(x) & am) || (reinterpret_cast<uintptr_t>(y) & am) ||
            (reinterpret_cast<uintptr_t>(argmax) & 3))
            return MR_ERR_BADARG;
        p.argmax = argmax;
        const dim3 g((unsigned)sp_nhwc_blocks((int64_t)batch_size * p.OH * p.OW, channels));
        if (act_dtype == 0) hipLaunchKernelGGL(stem_pool_nhwc_forward_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(stem_pool_nhwc_forward_kernel<bf16_t>, g, dim3(256), 0, (hipStream_t)stream, p);
        MR_CHECK_LAUNCH();
        return MR_OK;
    }
    const dim3 grid((unsigned)((int64_t)batch_size * channels * p.tiles_x * p.tiles_y));
    if (act_dtype == 0) hipLaunchKernelGGL(stem_pool_forward_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(stem_pool_forward_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}

extern "C" int64_t mr_stem_pool_backward_workspace_bytes(int batch_size, int channels, int height, int width) {
    using namespace mr;
    StemParams p{};
    if (stem_fill(p, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f, batch_size, channels, height, width) != MR_OK) return -1;
    const int64_t nchw = (int64_t)batch_size * p.tiles_x * p.tiles_y;
    return (int64_t)2 * channels * (nchw > SP_NHWC_BLOCKS ? nchw : SP_NHWC_BLOCKS) * 4 + 16;
}

extern "C" int mr_stem_pool_backward(const void* grad_y, const void* grad_y2, const void* x, const unsigned char* argmax,
                                     const float* weight, const float* bias, const float* running_mean, const float* running_var, float eps,
                                     int act_dtype, int channels_last, void* grad_x, float* grad_weight,
                                     float* grad_bias, void* workspace, int64_t workspace_bytes, int batch_size,
                                     int channels, int height, int width, mr_stream_t stream) {
    using namespace mr;
    if (act_dtype != 0 && act_dtype != 1) return MR_ERR_BADARG;
    if (channels_last && channels > 0 && !sp_nhwc_ok(channels)) return MR_ERR_BADARG;
    StemParams p{};
    const int rc = stem_fill(p, x, weight, bias, running_mean, running_var, eps, batch_size, channels, height, width);
    if (rc != MR_OK) return rc;
    if (channels == 0) return MR_OK;
    const bool want_params = grad_weight || grad_bias;
    if (batch_size == 0 || height == 0 || width == 0) {
        hipError_t e = hipSuccess;
        if (grad_weight) e = hipMemsetAsync(grad_weight, 0, (size_t)channels * 4, (hipStream_t)stream);
        if (e == hipSuccess && grad_bias) e = hipMemsetAsync(grad_bias, 0, (size_t)channels * 4, (hipStream_t)stream);
        return e == hipSuccess ? MR_OK : (int)e;
    }
    if (!grad_y || !x || !weight || !bias || !running_mean || !running_var || !grad_x) return MR_ERR_BADARG;
    if (want_params &&
        (!workspace || workspace_bytes < mr_stem_pool_backward_workspace_bytes(batch_size, channels, height, width)))
        return MR_ERR_BADARG;
    p.grad_y = grad_y; p.grad_y2 = grad_y2; p.grad_x = grad_x;
    p.partial = want_params ? static_cast<float*>(workspace) : nullptr;
    const int tiles = p.tiles_x * p.tiles_y;
    int64_t slots = (int64_t)batch_size * tiles;
    if (channels_last) {
        const uintptr_t am = (uintptr_t)(act_dtype == 0 ? 15 : 7);
        if (!argmax || (reinterpret_cast<uintptr_t>(x) & am) || (reinterpret_cast<uintptr_t>(grad_y) & am) ||
            (reinterpret_cast<uintptr_t>(grad_y2) & am) ||
            (reinterpret_cast<uintptr_t>(grad_x) & am) || (reinterpret_cast<uintptr_t>(argmax) & 3))
            return MR_ERR_BADARG;
        p.argmax = const_cast<unsigned char*>(argmax);
        slots = sp_nhwc_blocks((int64_t)batch_size * height * width, channels);
        const dim3 g((unsigned)slots);
        if (act_dtype == 0) hipLaunchKernelGGL(stem_pool_nhwc_backward_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(stem_pool_nhwc_backward_kernel<bf16_t>, g, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        const dim3 grid((unsigned)((int64_t)batch_size * channels * tiles));
        if (act_dtype == 0) hipLaunchKernelGGL(stem_pool_backward_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(stem_pool_backward_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    MR_CHECK_LAUNCH();
    if (want_params) {
        hipLaunchKernelGGL(stem_finish_kernel, dim3((unsigned)channels), dim3(256), 0, (hipStream_t)stream, p.partial,
                           running_var, eps, grad_weight, grad_bias, channels, slots);
        MR_CHECK_LAUNCH();
    }
    return MR_OK;
}
