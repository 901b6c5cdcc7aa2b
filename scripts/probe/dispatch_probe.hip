// How fast does an MI355X start (and retire) workgroups that have next to nothing to do?  The floor under every
// "most workgroups find nothing and leave" launch of this repository (forward tile kernel before the tile list, the
// pair-loss kernels on frames that are 5/6 background).   hipcc --offload-arch=gfx950 -O3 dispatch_probe.hip -o /tmp/dp && /tmp/dp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_exit(const int* flags, int* out) {}
__global__ void k_load(const int* flags, int* out) {  // one scalar load, a dependent second one, then leave
    const int a = flags[blockIdx.x & 1023];
    if (flags[(a + blockIdx.x) & 1023] == 12345) out[blockIdx.x] = 1;
}
__global__ void k_load_lds(const int* flags, int* out) {  // the same with 64 B of LDS and a barrier resource
    __shared__ int s[16];
    const int a = flags[blockIdx.x & 1023];
    if (flags[(a + blockIdx.x) & 1023] == 12345) { s[threadIdx.x & 15] = a; __syncthreads(); out[blockIdx.x] = s[0]; }
}

// ... and with the register footprint of the pair-loss kernels (~80 VGPRs, in a branch that is never taken): does a
// wave that leaves at once cost more to start when it reserves more registers?
__global__ void __launch_bounds__(256) k_load_regs(const int* flags, int* out) {
    const int a = flags[blockIdx.x & 1023];
    if (flags[(a + blockIdx.x) & 1023] == 12345) {
        float v[72];
#pragma unroll
        for (int i = 0; i < 72; i++) v[i] = __int_as_float(flags[(threadIdx.x + i * 7) & 1023]);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 72; i++) v[i] = v[i] * v[(i + 1) % 72] + v[(i + 5) % 72];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 72; i++) s += v[i];
        out[blockIdx.x * 256 + threadIdx.x] = __float_as_int(s);
    }
}

template <typename K>
static float time_us(K k, int grid, int block, const int* flags, int* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, flags, out);
    hipEventRecord(a, 0);
    const int n = 20;
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, flags, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / n;
}

int main() {
    int *flags, *out;
    hipMalloc(&flags, 1024 * 4); hipMemset(flags, 0, 1024 * 4);
    hipMalloc(&out, (size_t)65536 * 256 * 4);
    const int grids[] = {2048, 4096, 16384, 32768, 65536};
    const int blocks[] = {64, 256, 1024};
    printf("us per launch (20 back-to-back launches):\n%8s %6s %10s %10s %10s %12s\n", "grid", "block", "exit", "2 loads", "2 loads+LDS", "2 loads+regs");
    for (int g : grids)
        for (int b : blocks)
            printf("%8d %6d %10.1f %10.1f %10.1f %12.1f\n", g, b, time_us(k_exit, g, b, flags, out), time_us(k_load, g, b, flags, out),
                   time_us(k_load_lds, g, b, flags, out), b <= 256 ? time_us(k_load_regs, g, b, flags, out) : 0.f);
    return 0;
}
