// Micro-benchmark: sustained issue rate of the f32 MFMA shapes on gfx950 (cycles per instruction at 2.4 GHz, TFLOP/s), for
// 1, 2 and 4 waves per SIMD.   build: hipcc -O3 --offload-arch=gfx950 -o mfma_shapes tools/micro/mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE, int W>   // SHAPE 0: 32x32x2 (16 acc regs), 1: 16x16x4 (4 acc regs); W waves per SIMD
__global__ void __launch_bounds__(256 * W)
k_shape(int iters, float* out) {
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    float s = 0;
    if (SHAPE == 0) {
        f16v a[8];
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 16; ++e) a[k][e] = 0.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[k & 7], 0, 0, 0);
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 16; ++e) s += a[k][e];
    } else {
        f4v a[8];
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 4; ++e) a[k][e] = 0.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[k & 7], 0, 0, 0);
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 4; ++e) s += a[k][e];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int SHAPE, int W>
static void run(float* out) {
    const int waves_per_simd = W;
    const int iters = 20000 / waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_shape<SHAPE, W>), dim3(256), dim3(256 * waves_per_simd), 0, 0, iters, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_shape<SHAPE, W>), dim3(256), dim3(256 * waves_per_simd), 0, 0, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * (SHAPE == 0 ? 16 : 32) * waves_per_simd;     // MFMAs issued on one SIMD
    const double flop = per_simd * 1024.0 * (SHAPE == 0 ? 4096.0 : 2048.0);
    printf("%-10s %d wave(s)/SIMD  %8.3f ms  %6.2f cycles per MFMA (nominal %d)  %6.1f TFLOP/s\n", SHAPE == 0 ? "32x32x2" : "16x16x4",
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / per_simd, SHAPE == 0 ? 64 : 32, flop / ms * 1e-9);
}

int main() {
    float* out; (void)hipMalloc(&out, 4096 * 4);
    run<0, 1>(out); run<0, 2>(out); run<0, 3>(out);
    run<1, 1>(out); run<1, 2>(out); run<1, 4>(out);
    return 0;
}
