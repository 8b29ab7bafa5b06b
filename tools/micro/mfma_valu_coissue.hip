// Micro-benchmark: can a SIMD of gfx950 run f32 MFMA (v_mfma_f32_32x32x2_f32) from two waves and packed f32 FMA
// (v_pk_fma_f32) from a third wave at the same time, and at what combined rate?
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/coissue tools/micro/mfma_valu_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

// waves [0, nm) of each workgroup run MFMA, waves [nm, nm+nv) run pk_fma.  iters: MFMA waves do iters*4 MFMAs (4 accumulators),
// VALU waves do iters*vmul*16 pk_fma.
__global__ void __launch_bounds__(1024)
k_mix(int nm, int nv, int iters, int vmul, float* out) {
    const int wv = threadIdx.x >> 6;
    if (wv < nm) {
        f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
        float s = 0;
        for (int k = 0; k < 16; ++k) s += a0[k] + a1[k] + a2[k] + a3[k];
        if (s == 12345.678f) out[threadIdx.x] = s;
    } else if (wv < nm + nv) {
        f2v acc[16];
        for (int k = 0; k < 16; ++k) acc[k] = f2v{(float)k, (float)threadIdx.x};
        f2v z = {1.0001f, 0.9999f}, w = {0.5f, 0.25f};
        for (int i = 0; i < iters * vmul; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(z), "v"(w));
        }
        float s = 0;
        for (int k = 0; k < 16; ++k) s += acc[k].x + acc[k].y;
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
}

static double run(int nm, int nv, int iters, int vmul, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 64 * (nm + nv);
    hipLaunchKernelGGL(k_mix, dim3(256), dim3(threads), 0, 0, nm, nv, iters, vmul, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix, dim3(256), dim3(threads), 0, 0, nm, nv, iters, vmul, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 4096 * 4);
    const int iters = 20000;
    // one workgroup per CU (256 blocks); waves round-robin over the 4 SIMDs
    struct { int nm, nv, vmul; const char* what; } cfg[] = {
        {8, 0, 0, "MFMA only, 2 waves/SIMD"},
        {4, 0, 0, "MFMA only, 1 wave/SIMD"},
        {0, 4, 4, "pk_fma only, 1 wave/SIMD (same FLOPs as 4 MFMA waves x1)"},
        {0, 8, 4, "pk_fma only, 2 waves/SIMD"},
        {8, 4, 8, "MFMA 2 waves/SIMD + pk_fma 1 wave/SIMD (equal FLOPs per pipe)"},
        {8, 4, 4, "MFMA 2 waves/SIMD + pk_fma 1 wave/SIMD (VALU half the FLOPs)"},
        {8, 8, 4, "MFMA 2 waves/SIMD + pk_fma 2 waves/SIMD (equal FLOPs)"},
        {4, 4, 4, "MFMA 1 wave/SIMD + pk_fma 1 wave/SIMD (equal FLOPs)"},
    };
    for (auto& c : cfg) {
        const double ms = run(c.nm, c.nv, iters, c.vmul, out);
        // FLOPs: MFMA wave: iters*4 * 4096; VALU wave: iters*vmul*16 * 64 lanes * 4
        const double fm = 256.0 * c.nm * iters * 4.0 * 4096.0, fv = 256.0 * c.nv * (double)iters * c.vmul * 16.0 * 256.0;
        printf("%-70s %8.3f ms  mfma %7.1f TF  valu %7.1f TF  total %7.1f TF\n", c.what, ms, fm / ms * 1e-9, fv / ms * 1e-9,
               (fm + fv) / ms * 1e-9);
    }
    return 0;
}
