// frame_batch.hip -- decoded frames -> network-input batch on gfx950 (SURVEY 8f "f4").
//
// Replaces, for a whole batch of frames in one launch, what the reference does per sample in its
// DataLoader workers (meshreg/datasets/handobjset.py:361-379):
//   img = transform_img(img, affinetrans, inp_res)   libyana -> PIL Image.transform(AFFINE), NEAREST, fill 0
//   img = img.crop((0, 0, W, H)); to_tensor(img) ; normalize(img, mean, std)
//   jittermask = to_tensor(transform_img(white image, affinetrans, inp_res))
// (and the optional left-right flip of handobjset.py:124-125).
//
// Byte work, HBM-bound: per output pixel 3 source bytes in, 12 B of image + 4 or 12 B of mask out.
// Bit-exact with Pillow: the source pixel of every output pixel is computed exactly the way Pillow's
// Geometry.c does it, in each of its three regimes --
//   scale      (b == 0 && d == 0)  positions ACCUMULATED in double along each axis, one table per axis
//                                  (what block_rot / max_rot=0 training uses);
//   fixed      16.16 fixed point with 32-bit wrap-around, closed form per pixel (rotations);
//   double     positions accumulated in double along y then x (coefficients beyond +-32768).
// The accumulations are order-dependent, so a tiny per-frame pre-kernel walks them sequentially (<= W + H
// dependent adds) and the streaming kernel only looks the tables up.
#include "mr_common.hpp"

namespace mr {

enum { FB_SCALE = 0, FB_FIXED = 1, FB_DOUBLE = 2, FB_OUTSIDE = 3 };

struct __attribute__((aligned(8))) FrameHdr {
    int mode;
    int fix[6];  // 16.16 coefficients a0 a1 a2 a3 a4 a5 (FB_FIXED)
    int pad;
};

struct FrameBatchParams {
    const uint8_t* frames;  // [N,Hs,Ws,3]
    const double* coeffs;   // [N,6]  (a b c d e f): output (x,y) <- input (a x + b y + c, d x + e y + f)
    const uint8_t* flip;    // [N] or NULL
    float mean[3], stdv[3];
    FrameHdr* hdr;          // [N]
    double* rowx;           // [N,H]   FB_DOUBLE: position of pixel 0 of every row
    double* rowy;           // [N,H]
    int* xtab;              // [N,W]   FB_SCALE: source column (or -1)
    int* ytab;              // [N,H]
    float* image;           // [N,3,H,W]
    float* mask;            // [N,mc,H,W] or NULL
    int mask_channels;
    int N, Hs, Ws, H, W;
};

// Pillow's COORD(): (int)v for v >= 0, -1 below; out-of-int-range and NaN land outside as well
__device__ __forceinline__ int coord(double v) {
    return (v >= 0.0 && v < 2147483648.0) ? (int)v : -1;
}

// Pillow's FIX(): floor(v * 65536 + 0.5) as a 32-bit int (low 32 bits when out of range)
__device__ __forceinline__ int fix16(double v) {
    const double f = floor(v * 65536.0 + 0.5);
    if (!(f > -9.0e18 && f < 9.0e18)) return 0;
    return (int)(unsigned)(unsigned long long)(long long)f;
}

__device__ __forceinline__ bool fits_fixed(const double* a, double x, double y) {
    return fabs(x * a[0] + y * a[1] + a[2]) < 32768.0 && fabs(x * a[3] + y * a[4] + a[5]) < 32768.0;
}

// one block per frame; the sequential walks are split over a few lanes
__global__ void frame_tables_kernel(FrameBatchParams p) {
    const int n = blockIdx.x;
    const double* a = p.coeffs + 6 * (size_t)n;
    bool finite = true;
    for (int k = 0; k < 6; k++) finite = finite && isfinite(a[k]);
    int mode;
    if (!finite) mode = FB_OUTSIDE;
    else if (a[1] == 0.0 && a[3] == 0.0) mode = FB_SCALE;
    else if (fits_fixed(a, 0, 0) && fits_fixed(a, p.W, p.H) && fits_fixed(a, 0, p.H) && fits_fixed(a, p.W, 0)) mode = FB_FIXED;
    else mode = FB_DOUBLE;
    const int lane = threadIdx.x;
    if (lane == 0) {
        FrameHdr h;
        h.mode = mode;
        h.fix[0] = fix16(a[0]); h.fix[1] = fix16(a[1]); h.fix[3] = fix16(a[3]); h.fix[4] = fix16(a[4]);
        h.fix[2] = fix16(a[2] + a[0] * 0.5 + a[1] * 0.5);
        h.fix[5] = fix16(a[5] + a[3] * 0.5 + a[4] * 0.5);
        h.pad = 0;
        p.hdr[n] = h;
    }
    if (mode == FB_SCALE) {
        if (lane == 0) {
            int* xt = p.xtab + (size_t)n * p.W;
            double xo = a[2] + a[0] * 0.5;
            for (int x = 0; x < p.W; x++) {
                const int xin = coord(xo);
                xt[x] = (xin >= 0 && xin < p.Ws) ? xin : -1;
                xo += a[0];
            }
        } else if (lane == 1) {
            int* yt = p.ytab + (size_t)n * p.H;
            double yo = a[5] + a[4] * 0.5;
            for (int y = 0; y < p.H; y++) {
                const int yin = coord(yo);
                yt[y] = (yin >= 0 && yin < p.Hs) ? yin : -1;
                yo += a[4];
            }
        }
    } else if (mode == FB_DOUBLE) {
        if (lane == 0) {
            double* rx = p.rowx + (size_t)n * p.H;
            double xx = a[2] + a[0] * 0.5 + a[1] * 0.5;
            for (int y = 0; y < p.H; y++) { rx[y] = xx; xx += a[1]; }
        } else if (lane == 1) {
            double* ry = p.rowy + (size_t)n * p.H;
            double yy = a[5] + a[3] * 0.5 + a[4] * 0.5;
            for (int y = 0; y < p.H; y++) { ry[y] = yy; yy += a[4]; }
        }
    }
}

constexpr int FB_PX = 4;  // output pixels per thread (one 16-byte store per plane)

// grid (ceil(W / (64*4)), ceil(H / 4), N), block (64, 4): a wave covers 256 consecutive pixels of one row
__global__ __launch_bounds__(256) void frames_to_batch_kernel(FrameBatchParams p) {
    const int n = blockIdx.z;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * FB_PX;
    if (y >= p.H || x0 >= p.W) return;
    const FrameHdr h = p.hdr[n];
    const bool flip = p.flip && p.flip[n];
    const uint8_t* src = p.frames + (size_t)n * p.Hs * p.Ws * 3;
    int xin[FB_PX], yin[FB_PX];
    if (h.mode == FB_SCALE) {
        const int yy = p.ytab[(size_t)n * p.H + y];
        const int* xt = p.xtab + (size_t)n * p.W;
#pragma unroll
        for (int i = 0; i < FB_PX; i++) {
            xin[i] = (x0 + i < p.W) ? xt[x0 + i] : -1;
            yin[i] = yy;
        }
    } else if (h.mode == FB_FIXED) {
        const unsigned bx = (unsigned)h.fix[2] + (unsigned)y * (unsigned)h.fix[1];
        const unsigned by = (unsigned)h.fix[5] + (unsigned)y * (unsigned)h.fix[4];
#pragma unroll
        for (int i = 0; i < FB_PX; i++) {
            xin[i] = (int)(bx + (unsigned)(x0 + i) * (unsigned)h.fix[0]) >> 16;
            yin[i] = (int)(by + (unsigned)(x0 + i) * (unsigned)h.fix[3]) >> 16;
        }
    } else if (h.mode == FB_DOUBLE) {
        const double* a = p.coeffs + 6 * (size_t)n;
        double xx = p.rowx[(size_t)n * p.H + y], yy = p.rowy[(size_t)n * p.H + y];
        for (int x = 0; x < x0; x++) { xx += a[0]; yy += a[3]; }  // Pillow's running sums along the row
#pragma unroll
        for (int i = 0; i < FB_PX; i++) {
            xin[i] = coord(xx);
            yin[i] = coord(yy);
            xx += a[0];
            yy += a[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < FB_PX; i++) xin[i] = yin[i] = -1;
    }
    float px[3][FB_PX], mk[FB_PX];
#pragma unroll
    for (int i = 0; i < FB_PX; i++) {
        const bool in = xin[i] >= 0 && xin[i] < p.Ws && yin[i] >= 0 && yin[i] < p.Hs;
        unsigned char c[3] = {0, 0, 0};
        if (in) {
            const int xs = flip ? p.Ws - 1 - xin[i] : xin[i];
            const uint8_t* s = src + ((size_t)yin[i] * p.Ws + xs) * 3;
            c[0] = s[0]; c[1] = s[1]; c[2] = s[2];
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) px[ch][i] = ((float)c[ch] / 255.0f - p.mean[ch]) / p.stdv[ch];
        mk[i] = in ? 1.0f : 0.0f;
    }
    const size_t plane = (size_t)p.H * p.W;
    const size_t o = (size_t)y * p.W + x0;
    const bool vec = (p.W % FB_PX) == 0;  // rows and planes stay 16-byte aligned
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float* dst = p.image + ((size_t)n * 3 + ch) * plane + o;
        if (vec) {
            *reinterpret_cast<float4*>(dst) = make_float4(px[ch][0], px[ch][1], px[ch][2], px[ch][3]);
        } else {
#pragma unroll
            for (int i = 0; i < FB_PX; i++)
                if (x0 + i < p.W) dst[i] = px[ch][i];
        }
    }
    if (p.mask) {
        for (int ch = 0; ch < p.mask_channels; ch++) {
            float* dst = p.mask + ((size_t)n * p.mask_channels + ch) * plane + o;
            if (vec) {
                *reinterpret_cast<float4*>(dst) = make_float4(mk[0], mk[1], mk[2], mk[3]);
            } else {
#pragma unroll
                for (int i = 0; i < FB_PX; i++)
                    if (x0 + i < p.W) dst[i] = mk[i];
            }
        }
    }
}

static inline int64_t fb_align(int64_t v) { return (v + 15) & ~(int64_t)15; }

}  // namespace mr

extern "C" int64_t mr_frames_to_batch_workspace_bytes(int num_frames, int height, int width) {
    if (num_frames < 0 || height < 0 || width < 0) return -1;
    const int64_t n = num_frames, H = height, W = width;
    return mr::fb_align(n * (int64_t)sizeof(mr::FrameHdr)) + 2 * mr::fb_align(n * H * 8) + mr::fb_align(n * W * 4) +
           mr::fb_align(n * H * 4);
}

extern "C" int mr_frames_to_batch(const uint8_t* frames, const double* coeffs, const uint8_t* flip, float mean0,
                                  float mean1, float mean2, float std0, float std1, float std2, void* workspace,
                                  int64_t workspace_bytes, float* image, float* jittermask, int mask_channels,
                                  int num_frames, int src_height, int src_width, int height, int width,
                                  mr_stream_t stream) {
    using namespace mr;
    if (num_frames < 0 || src_height < 0 || src_width < 0 || height < 0 || width < 0) return MR_ERR_BADARG;
    if (jittermask && mask_channels != 1 && mask_channels != 3) return MR_ERR_BADARG;
    if (num_frames == 0 || height == 0 || width == 0) return MR_OK;
    if (!frames && src_height > 0 && src_width > 0) return MR_ERR_BADARG;
    if (!coeffs || !image || !workspace) return MR_ERR_BADARG;
    if (num_frames > 65535 || height > 32767 || width > 32767 || src_height > 32767 || src_width > 32767)
        return MR_ERR_BADARG;
    if (workspace_bytes < mr_frames_to_batch_workspace_bytes(num_frames, height, width)) return MR_ERR_BADARG;
    if ((reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(image) & 15) ||
        (reinterpret_cast<uintptr_t>(jittermask) & 15))
        return MR_ERR_BADARG;
    FrameBatchParams p;
    p.frames = frames; p.coeffs = coeffs; p.flip = flip;
    p.mean[0] = mean0; p.mean[1] = mean1; p.mean[2] = mean2;
    p.stdv[0] = std0; p.stdv[1] = std1; p.stdv[2] = std2;
    char* w = static_cast<char*>(workspace);
    const int64_t n = num_frames;
    p.hdr = reinterpret_cast<FrameHdr*>(w); w += fb_align(n * (int64_t)sizeof(FrameHdr));
    p.rowx = reinterpret_cast<double*>(w);  w += fb_align(n * height * 8);
    p.rowy = reinterpret_cast<double*>(w);  w += fb_align(n * height * 8);
    p.xtab = reinterpret_cast<int*>(w);     w += fb_align(n * width * 4);
    p.ytab = reinterpret_cast<int*>(w);
    p.image = image; p.mask = jittermask; p.mask_channels = jittermask ? mask_channels : 0;
    p.N = num_frames; p.Hs = src_height; p.Ws = src_width; p.H = height; p.W = width;
    hipLaunchKernelGGL(frame_tables_kernel, dim3((unsigned)num_frames), dim3(64), 0, (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    const dim3 block(64, 4);
    const dim3 grid((unsigned)((width + 64 * FB_PX - 1) / (64 * FB_PX)), (unsigned)((height + 3) / 4), (unsigned)num_frames);
    hipLaunchKernelGGL(frames_to_batch_kernel, grid, block, 0, (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}
