// Streaming-read / -write probes: what a plain, well-formed kernel reaches on this GPU for a given footprint,
// cold (after a 768 MB flush) and warm.  Built and run by scripts/bw_probe.py; not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int UNROLL>
__global__ void __launch_bounds__(256) read_kernel(const float4* __restrict__ x, int64_t n4, float* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) { const float4 v = x[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void __launch_bounds__(256) write_kernel(float4* __restrict__ y, int64_t n4, float v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) y[i] = make_float4(v, v, v, v);
}

extern "C" int probe_read(const void* x, int64_t bytes, void* out, int blocks, int unroll, void* stream) {
    const int64_t n4 = bytes / 16;
    if (unroll == 8) hipLaunchKernelGGL(read_kernel<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, n4, (float*)out);
    else if (unroll == 4) hipLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, n4, (float*)out);
    else hipLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, n4, (float*)out);
    return (int)hipGetLastError();
}
extern "C" int probe_write(void* y, int64_t bytes, int blocks, void* stream) {
    hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)y, bytes / 16, 1.0f);
    return (int)hipGetLastError();
}
