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
    float* lut;             // [3,256] (u8 / 255 - mean) / std per channel
    double* rowx;           // [N,H]   FB_DOUBLE: position of pixel 0 of every row
    double* rowy;           // [N,H]
    int* xtab;              // [N,W]   FB_SCALE: source column (or -1)
    int* ytab;              // [N,H]
    float* image;           // [N,3,H,W]
    float* mask;            // [N,mc,H,W] or NULL
    int mask_channels;
    int N, Hs, Ws, H, W;
    int Hs_ld, Ws_ld;       // max(Hs, 1), max(Ws, 1): clamp range of the unconditional source loads
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

constexpr int FB_CHUNK = 1024;  // positions accumulated per pass of the table kernel

// One block of two waves per frame: wave 0 walks x, wave 1 walks y.  Only the running double-precision sums are
// sequential (lane 0 of each wave, into LDS, one dependent add per position); truncation, range checks and the
// stores are done by all lanes.  Block 0 also fills the 3 x 256 table of normalised pixel values
// (u8 / 255 - mean) / std, so that the streaming kernel does no divisions.
__global__ __launch_bounds__(128) void frame_tables_kernel(FrameBatchParams p) {
    __shared__ double pos[2][FB_CHUNK];
    const int n = blockIdx.x;
    const double* a = p.coeffs + 6 * (size_t)n;
    bool finite = true;
    for (int k = 0; k < 6; k++) finite = finite && isfinite(a[k]);
    int mode;
    if (!finite) mode = FB_OUTSIDE;
    else if (a[1] == 0.0 && a[3] == 0.0) mode = FB_SCALE;
    else if (fits_fixed(a, 0, 0) && fits_fixed(a, p.W, p.H) && fits_fixed(a, 0, p.H) && fits_fixed(a, p.W, 0)) mode = FB_FIXED;
    else mode = FB_DOUBLE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        FrameHdr h;
        h.mode = mode;
        h.fix[0] = fix16(a[0]); h.fix[1] = fix16(a[1]); h.fix[3] = fix16(a[3]); h.fix[4] = fix16(a[4]);
        h.fix[2] = fix16(a[2] + a[0] * 0.5 + a[1] * 0.5);
        h.fix[5] = fix16(a[5] + a[3] * 0.5 + a[4] * 0.5);
        h.pad = 0;
        p.hdr[n] = h;
    }
    if (n == 0)
        for (int e = threadIdx.x; e < 768; e += blockDim.x) {
            const int ch = e >> 8;
            p.lut[e] = ((float)(e & 255) / 255.0f - p.mean[ch]) / p.stdv[ch];
        }
    if (mode != FB_SCALE && mode != FB_DOUBLE) return;
    // wave 0: along x (scale) / row starts in x (double); wave 1: along y / row starts in y
    const int count = (mode == FB_SCALE && wave == 0) ? p.W : p.H;
    double start, step;
    if (mode == FB_SCALE) {
        start = wave == 0 ? a[2] + a[0] * 0.5 : a[5] + a[4] * 0.5;
        step = wave == 0 ? a[0] : a[4];
    } else {
        start = wave == 0 ? a[2] + a[0] * 0.5 + a[1] * 0.5 : a[5] + a[3] * 0.5 + a[4] * 0.5;
        step = wave == 0 ? a[1] : a[4];
    }
    const int limit = wave == 0 ? p.Ws : p.Hs;
    double run = start;
    for (int base = 0; base < count; base += FB_CHUNK) {
        const int m = min(FB_CHUNK, count - base);
        if (lane == 0)
            for (int k = 0; k < m; k++) {
                pos[wave][k] = run;
                run += step;
            }
        __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave execute in program order)
        for (int k = lane; k < m; k += 64) {
            const double v = pos[wave][k];
            if (mode == FB_SCALE) {
                const int c = coord(v);
                int* tab = wave == 0 ? p.xtab + (size_t)n * p.W : p.ytab + (size_t)n * p.H;
                tab[base + k] = (c >= 0 && c < limit) ? c : -1;
            } else {
                double* row = wave == 0 ? p.rowx + (size_t)n * p.H : p.rowy + (size_t)n * p.H;
                row[base + k] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// write-once output: streaming (non-temporal) 16-byte stores keep the source lines in L2 (cache-warm sources:
// 63 us instead of 106 for the 3 x 64 frames of a step; 302 MB written)
typedef float fb_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4_stream(float* dst, float a, float b, float c, float d) {
    fb_f4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<fb_f4*>(dst));
}

constexpr int FB_PX = 4;    // output pixels per thread and row (one 16-byte store per plane)
constexpr int FB_ROWS = 4;  // rows per thread: 16 source pixels requested before the first is used (B=64 bench shape, inputs cold:
                            // 100 us vs 107 with one row; putting all tiles of a frame on one XCD measured slower: 128 us)

// Source position of the FB_PX pixels starting at (x0, y); xt = the thread's slice of the x table (scale regime).
__device__ __forceinline__ void source_pos(const FrameBatchParams& p, const FrameHdr& h, int n, int x0, int y,
                                           const int* xt, int* xin, int* yin) {
    if (h.mode == FB_SCALE) {
        const int yy = p.ytab[(size_t)n * p.H + y];
#pragma unroll
        for (int i = 0; i < FB_PX; i++) {
            xin[i] = xt[i];
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
}

// grid (ceil(W / 256), ceil(H / (4 FB_ROWS)), N), block (64, 4): a wave covers 256 consecutive pixels of rows y0 + ty, y0 + ty + 4, ...  Per thread: the x-table slice once, then ALL source bytes of its FB_ROWS x FB_PX pixels
// are requested -- unconditionally, from clamped addresses -- before the first is used, then the stores.
__global__ __launch_bounds__(256) void frames_to_batch_kernel(FrameBatchParams p) {
    __shared__ float lut[768];
    for (int e = threadIdx.y * 64 + threadIdx.x; e < 768; e += 256) lut[e] = p.lut[e];
    const int n = blockIdx.z;
    const int ybase = blockIdx.y * (4 * FB_ROWS) + threadIdx.y;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * FB_PX;
    const FrameHdr h = p.hdr[n];
    const bool flip = p.flip && p.flip[n];
    // (empty source: Hs * Ws == 0; the host points `frames` at readable memory and Hs_ld = Ws_ld = 1)
    const uint8_t* src = p.frames + (size_t)n * p.Hs * p.Ws * 3;
    int xt[FB_PX];
#pragma unroll
    for (int i = 0; i < FB_PX; i++) xt[i] = -1;
    if (h.mode == FB_SCALE) {
        const int* tab = p.xtab + (size_t)n * p.W;
#pragma unroll
        for (int i = 0; i < FB_PX; i++) {
            const int xv = tab[min(x0 + i, p.W - 1)];
            xt[i] = (x0 + i < p.W) ? xv : -1;
        }
    }
    bool in[FB_ROWS][FB_PX];
    unsigned c[FB_ROWS][FB_PX][3];
#pragma unroll
    for (int r = 0; r < FB_ROWS; r++) {
        const int y = min(ybase + 4 * r, p.H - 1);
        int xin[FB_PX], yin[FB_PX];
        source_pos(p, h, n, min(x0, p.W - 1), y, xt, xin, yin);
#pragma unroll
        for (int i = 0; i < FB_PX; i++) {
            in[r][i] = xin[i] >= 0 && xin[i] < p.Ws && yin[i] >= 0 && yin[i] < p.Hs;  // never true for an empty source
            const int xc = min(max(xin[i], 0), p.Ws_ld - 1), yc = min(max(yin[i], 0), p.Hs_ld - 1);
            const int xs = flip ? p.Ws_ld - 1 - xc : xc;
            const uint8_t* s = src + ((size_t)yc * p.Ws_ld + xs) * 3;
            c[r][i][0] = s[0];
            c[r][i][1] = s[1];
            c[r][i][2] = s[2];
        }
    }
    __syncthreads();  // the lookup table is in LDS
    if (x0 >= p.W) return;
    const size_t plane = (size_t)p.H * p.W;
    const bool vec = (p.W % FB_PX) == 0;  // rows and planes stay 16-byte aligned
#pragma unroll
    for (int r = 0; r < FB_ROWS; r++) {
        const int y = ybase + 4 * r;
        if (y >= p.H) break;
        const size_t o = (size_t)y * p.W + x0;
        float mk[FB_PX];
#pragma unroll
        for (int i = 0; i < FB_PX; i++) mk[i] = in[r][i] ? 1.0f : 0.0f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float px[FB_PX];
#pragma unroll
            for (int i = 0; i < FB_PX; i++) px[i] = lut[ch * 256 + (in[r][i] ? c[r][i][ch] : 0u)];
            float* dst = p.image + ((size_t)n * 3 + ch) * plane + o;
            if (vec) {
                store4_stream(dst, px[0], px[1], px[2], px[3]);
            } else {
#pragma unroll
                for (int i = 0; i < FB_PX; i++)
                    if (x0 + i < p.W) dst[i] = px[i];
            }
        }
        if (p.mask) {
            for (int ch = 0; ch < p.mask_channels; ch++) {
                float* dst = p.mask + ((size_t)n * p.mask_channels + ch) * plane + o;
                if (vec) {
                    store4_stream(dst, mk[0], mk[1], mk[2], mk[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < FB_PX; i++)
                        if (x0 + i < p.W) dst[i] = mk[i];
                }
            }
        }
    }
}

static inline int64_t fb_align(int64_t v) { return (v + 15) & ~(int64_t)15; }

}  // namespace mr

extern "C" int64_t mr_frames_to_batch_workspace_bytes(int num_frames, int height, int width) {
    if (num_frames < 0 || height < 0 || width < 0) return -1;
    const int64_t n = num_frames, H = height, W = width;
    return mr::fb_align(768 * 4) + mr::fb_align(n * (int64_t)sizeof(mr::FrameHdr)) + 2 * mr::fb_align(n * H * 8) + mr::fb_align(n * W * 4) +
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
    p.lut = reinterpret_cast<float*>(w);    w += fb_align(768 * 4);
    p.hdr = reinterpret_cast<FrameHdr*>(w); w += fb_align(n * (int64_t)sizeof(FrameHdr));
    p.rowx = reinterpret_cast<double*>(w);  w += fb_align(n * height * 8);
    p.rowy = reinterpret_cast<double*>(w);  w += fb_align(n * height * 8);
    p.xtab = reinterpret_cast<int*>(w);     w += fb_align(n * width * 4);
    p.ytab = reinterpret_cast<int*>(w);
    p.image = image; p.mask = jittermask; p.mask_channels = jittermask ? mask_channels : 0;
    p.N = num_frames; p.Hs = src_height; p.Ws = src_width; p.H = height; p.W = width;
    p.Hs_ld = src_height > 0 ? src_height : 1;
    p.Ws_ld = src_width > 0 ? src_width : 1;
    if (src_height == 0 || src_width == 0) p.frames = static_cast<const uint8_t*>(workspace);  // readable, never selected
    hipLaunchKernelGGL(frame_tables_kernel, dim3((unsigned)num_frames), dim3(128), 0, (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    const dim3 block(64, 4);
    const dim3 grid((unsigned)((width + 64 * FB_PX - 1) / (64 * FB_PX)), (unsigned)((height + 4 * FB_ROWS - 1) / (4 * FB_ROWS)),
                    (unsigned)num_frames);
    hipLaunchKernelGGL(frames_to_batch_kernel, grid, block, 0, (hipStream_t)stream, p);
    MR_CHECK_LAUNCH();
    return MR_OK;
}
