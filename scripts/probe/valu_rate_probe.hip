// Issue cost of single instructions on gfx950, in cycles per wave-instruction on one SIMD: which of the operations the
// scatter / pixel-map kernels are built from are full-rate (4 cycles per wave64), which are not.
//   hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o /tmp/vr && /tmp/vr
// Method: one wave per SIMD slot (256 threads per workgroup, one workgroup per CU), each wave runs ITER trips of a loop
// whose body is 16 copies of the instruction on 8 independent register sets (no dependent chain shorter than 8);
// s_memtime around the loop, cycles = ticks (100 MHz wall clock) * clock ratio taken from a v_fma_f32 run of the same shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP16(X) X X X X X X X X X X X X X X X X
constexpr int ITER = 2000;

#define PROBE_F32(name, ASM)                                                                  \
    __global__ void __launch_bounds__(256) name(float* out, unsigned long long* t) {         \
        float a0 = threadIdx.x * 0.5f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;         \
        float a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0001f, c = 0.5f;     \
        const unsigned long long t0 = __builtin_readcyclecounter();                           \
        for (int i = 0; i < ITER; i++) {                                                      \
            asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                     \
        const unsigned long long t1 = __builtin_readcyclecounter();                           \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
        if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;                                        \
    }

// 16 instructions per trip: two rounds over the eight registers
#define TWO(X) X "\n" X "\n"
PROBE_F32(k_fma_f32, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9")
#define ONEOP(op) op " %0, %0\n " op " %1, %1\n " op " %2, %2\n " op " %3, %3\n " op " %4, %4\n " op " %5, %5\n " op " %6, %6\n " op " %7, %7\n " op " %0, %0\n " op " %1, %1\n " op " %2, %2\n " op " %3, %3\n " op " %4, %4\n " op " %5, %5\n " op " %6, %6\n " op " %7, %7"
PROBE_F32(k_rcp_f32, ONEOP("v_rcp_f32"))
PROBE_F32(k_trunc_f32, ONEOP("v_trunc_f32"))
PROBE_F32(k_cvt_u32_f32, ONEOP("v_cvt_u32_f32"))
PROBE_F32(k_cvt_f32_u32, ONEOP("v_cvt_f32_u32"))
#define TWOOP(op) op " %0, %0, %8\n " op " %1, %1, %8\n " op " %2, %2, %8\n " op " %3, %3, %8\n " op " %4, %4, %8\n " op " %5, %5, %8\n " op " %6, %6, %8\n " op " %7, %7, %8\n " op " %0, %0, %8\n " op " %1, %1, %8\n " op " %2, %2, %8\n " op " %3, %3, %8\n " op " %4, %4, %8\n " op " %5, %5, %8\n " op " %6, %6, %8\n " op " %7, %7, %8"
PROBE_F32(k_mul_f32, TWOOP("v_mul_f32"))
PROBE_F32(k_max_f32, TWOOP("v_max_f32"))
PROBE_F32(k_xor_b32, TWOOP("v_xor_b32"))
#define THREEOP(op) op " %0, %0, %8, %9\n " op " %1, %1, %8, %9\n " op " %2, %2, %8, %9\n " op " %3, %3, %8, %9\n " op " %4, %4, %8, %9\n " op " %5, %5, %8, %9\n " op " %6, %6, %8, %9\n " op " %7, %7, %8, %9\n " op " %0, %0, %8, %9\n " op " %1, %1, %8, %9\n " op " %2, %2, %8, %9\n " op " %3, %3, %8, %9\n " op " %4, %4, %8, %9\n " op " %5, %5, %8, %9\n " op " %6, %6, %8, %9\n " op " %7, %7, %8, %9"
PROBE_F32(k_med3_f32, THREEOP("v_med3_f32"))
PROBE_F32(k_bfi_b32, THREEOP("v_bfi_b32"))
PROBE_F32(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n v_readlane_b32 s24, %4, 11\n v_readlane_b32 s25, %5, 13\n v_readlane_b32 s26, %6, 15\n v_readlane_b32 s27, %7, 17\n v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n v_readlane_b32 s24, %4, 11\n v_readlane_b32 s25, %5, 13\n v_readlane_b32 s26, %6, 15\n v_readlane_b32 s27, %7, 17")

#define PROBE_F64(name, ASM)                                                                  \
    __global__ void __launch_bounds__(256) name(float* out, unsigned long long* t) {         \
        double a0 = threadIdx.x * 0.5 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;           \
        double a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0001, c = 0.5;       \
        int e = 1;                                                                            \
        const unsigned long long t0 = __builtin_readcyclecounter();                           \
        for (int i = 0; i < ITER; i++) {                                                      \
            asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(e)); \
        }                                                                                     \
        const unsigned long long t1 = __builtin_readcyclecounter();                           \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
        if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;                                        \
    }
PROBE_F64(k_fma_f64, THREEOP("v_fma_f64"))
PROBE_F64(k_trunc_f64, ONEOP("v_trunc_f64"))
PROBE_F64(k_floor_f64, ONEOP("v_floor_f64"))
#define LDEXP(op) op " %0, %0, %10\n " op " %1, %1, %10\n " op " %2, %2, %10\n " op " %3, %3, %10\n " op " %4, %4, %10\n " op " %5, %5, %10\n " op " %6, %6, %10\n " op " %7, %7, %10\n " op " %0, %0, %10\n " op " %1, %1, %10\n " op " %2, %2, %10\n " op " %3, %3, %10\n " op " %4, %4, %10\n " op " %5, %5, %10\n " op " %6, %6, %10\n " op " %7, %7, %10"
PROBE_F64(k_ldexp_f64, LDEXP("v_ldexp_f64"))
PROBE_F64(k_lshl_b64, "v_lshlrev_b64 %0, %10, %0\n v_lshlrev_b64 %1, %10, %1\n v_lshlrev_b64 %2, %10, %2\n v_lshlrev_b64 %3, %10, %3\n v_lshlrev_b64 %4, %10, %4\n v_lshlrev_b64 %5, %10, %5\n v_lshlrev_b64 %6, %10, %6\n v_lshlrev_b64 %7, %10, %7\n v_lshlrev_b64 %0, %10, %0\n v_lshlrev_b64 %1, %10, %1\n v_lshlrev_b64 %2, %10, %2\n v_lshlrev_b64 %3, %10, %3\n v_lshlrev_b64 %4, %10, %4\n v_lshlrev_b64 %5, %10, %5\n v_lshlrev_b64 %6, %10, %6\n v_lshlrev_b64 %7, %10, %7")
PROBE_F64(k_pk_fma_f32, THREEOP("v_pk_fma_f32"))
PROBE_F64(k_pk_mul_f32, TWOOP("v_pk_mul_f32"))
// conversions between widths: destination and source differ in size, use scratch registers
__global__ void __launch_bounds__(256) k_cvt_f64_f32(float* out, unsigned long long* t) {
    float a0 = threadIdx.x * 0.5f + 1.0f;
    double d0, d1, d2, d3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4\n v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4\n"
                     "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4\n v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
__global__ void __launch_bounds__(256) k_cvt_u32_f64(float* out, unsigned long long* t) {
    double a0 = threadIdx.x * 0.5 + 1.0;
    unsigned d0, d1, d2, d3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cvt_u32_f64 %0, %4\n v_cvt_u32_f64 %1, %4\n v_cvt_u32_f64 %2, %4\n v_cvt_u32_f64 %3, %4\n v_cvt_u32_f64 %0, %4\n v_cvt_u32_f64 %1, %4\n v_cvt_u32_f64 %2, %4\n v_cvt_u32_f64 %3, %4\n"
                     "v_cvt_u32_f64 %0, %4\n v_cvt_u32_f64 %1, %4\n v_cvt_u32_f64 %2, %4\n v_cvt_u32_f64 %3, %4\n v_cvt_u32_f64 %0, %4\n v_cvt_u32_f64 %1, %4\n v_cvt_u32_f64 %2, %4\n v_cvt_u32_f64 %3, %4"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
// LDS 64-bit atomics: every lane its own cell (stride 8 B) / all lanes of a wave 8 cells / one cell
template <int MODE>
__global__ void __launch_bounds__(256) k_ds_add_u64(float* out, unsigned long long* t) {
    __shared__ unsigned long long tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cell = wave * 512 + (MODE == 0 ? lane : MODE == 1 ? (lane & 7) * 2 : MODE == 2 ? 0 : (lane * 2) % 64 + (lane >> 5));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) atomicAdd(&tab[cell + ((k & 7) << 6)], (unsigned long long)(i + k + 1));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (float)tab[threadIdx.x];
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_ds_add_f32(float* out, unsigned long long* t) {
    __shared__ float tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cell = wave * 1024 + (MODE == 0 ? lane : MODE == 1 ? (lane & 7) * 2 : MODE == 2 ? 0 : (lane * 2) % 64 + (lane >> 5));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) unsafeAtomicAdd(&tab[cell + ((k & 7) << 6)], (float)(i + k + 1));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = tab[threadIdx.x];
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <typename K>
static double run(K k, int waves_per_simd, float* out, unsigned long long* t_dev, int cus) {
    hipMemset(t_dev, 0, 8 * 4096);
    const int grid = cus * waves_per_simd;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, t_dev);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, t_dev);
    hipDeviceSynchronize();
    static unsigned long long t[4096];
    hipMemcpy(t, t_dev, 8 * grid, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < grid; i++) s += (double)t[i];
    return s / grid / (ITER * 16.0);  // clock ticks per instruction of one wave
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out; unsigned long long* t;
    hipMalloc(&out, (size_t)4096 * 256 * 4); hipMalloc(&t, 8 * 4096);
    printf("CUs %d; ticks of __builtin_readcyclecounter per wave-instruction, 1 / 2 / 4 waves per SIMD (a full-rate VALU instruction = the v_fma_f32 row)\n", cus);
#define ROW(name, k) printf("%-22s %8.3f %8.3f %8.3f\n", name, run(k, 1, out, t, cus), run(k, 2, out, t, cus), run(k, 4, out, t, cus));
    ROW("v_fma_f32", k_fma_f32) ROW("v_mul_f32", k_mul_f32) ROW("v_max_f32", k_max_f32) ROW("v_xor_b32", k_xor_b32)
    ROW("v_med3_f32", k_med3_f32) ROW("v_bfi_b32", k_bfi_b32) ROW("v_rcp_f32", k_rcp_f32) ROW("v_trunc_f32", k_trunc_f32)
    ROW("v_cvt_u32_f32", k_cvt_u32_f32) ROW("v_cvt_f32_u32", k_cvt_f32_u32) ROW("v_readlane_b32", k_readlane)
    ROW("v_pk_fma_f32", k_pk_fma_f32) ROW("v_pk_mul_f32", k_pk_mul_f32)
    ROW("v_fma_f64", k_fma_f64) ROW("v_trunc_f64", k_trunc_f64) ROW("v_floor_f64", k_floor_f64) ROW("v_ldexp_f64", k_ldexp_f64)
    ROW("v_lshlrev_b64", k_lshl_b64) ROW("v_cvt_f64_f32", k_cvt_f64_f32) ROW("v_cvt_u32_f64", k_cvt_u32_f64)
    ROW("ds_add_u64 distinct", k_ds_add_u64<0>) ROW("ds_add_u64 8 cells", k_ds_add_u64<1>) ROW("ds_add_u64 one cell", k_ds_add_u64<2>)
    ROW("ds_add_u64 pairs", k_ds_add_u64<3>)
    ROW("ds_add_f32 distinct", k_ds_add_f32<0>) ROW("ds_add_f32 8 cells", k_ds_add_f32<1>) ROW("ds_add_f32 one cell", k_ds_add_f32<2>)
    ROW("ds_add_f32 pairs", k_ds_add_f32<3>)
    return 0;
}
