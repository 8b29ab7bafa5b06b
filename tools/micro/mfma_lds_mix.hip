// Micro-benchmark: what does an LDS read cost a wave that is streaming f32 MFMAs (v_mfma_f32_32x32x2_f32) on gfx950?
// One wave per SIMD runs 8 independent accumulators (as the pair kernel does); variants add ds_read traffic from the same wave,
// from a second wave on the same SIMD, or VALU work.   build: hipcc -O3 --offload-arch=gfx950 -o mfma_lds tools/micro/mfma_lds_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// mode: 0 MFMA only; 1 + one ds_read_b128 per 3 MFMAs (same wave); 2 + two ds_read_b64 per 3 MFMAs; 3 + one ds_read_b32 per 3;
//       4 + ds_read_b128 per 3 MFMAs, all lanes of a half-wave read the same address (broadcast);
//       5: waves 4-7 (second wave of each SIMD) do the ds_read_b128 stream, waves 0-3 MFMA only;  6: waves 4-7 do v_mul stream
//       7: + one ds_read_b128 per MFMA (same wave);  8 / 9: LDS float atomics (max, then min with return) from the same / a second wave
template <int MODE>
__global__ void __launch_bounds__(512)
k_mix(int iters, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 0.25f;
    __syncthreads();
    const unsigned base = (MODE == 4 ? (lane >> 5) * 256 : lane * 16) + wv * 2048;   // byte offset inside the array
    const unsigned addr = (unsigned)(size_t)lds + base;
    if (wv < 4) {
        f16v a[8];
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 16; ++e) a[k][e] = 0.f;
        float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
        f4v r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                a[k & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[k & 7], 0, 0, 0);
                if (MODE == 7 || ((MODE == 1 || MODE == 4) && k % 3 == 0)) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r0) : "v"(addr), "n"(0));
                } else if (MODE == 2 && k % 3 == 0) {
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=v"(*(double*)&r1), "=v"(*(double*)&r2) : "v"(addr));
                } else if (MODE == 3 && k % 3 == 0) {
                    asm volatile("ds_read_b32 %0, %1" : "=v"(r0.x) : "v"(addr));
                } else if (MODE == 8) {   // ReLU through the LDS atomic unit: max into a zeroed word, then min-with-return resets it
                    asm volatile("ds_max_f32 %1, %2\n\tds_min_rtn_f32 %0, %1, %3" : "=v"(r0.x) : "v"(addr), "v"(y), "v"(0.0f));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            x += r0.x * 1e-30f + r1.x * 1e-30f + r2.y * 1e-30f;
        }
        float s = 0;
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 16; ++e) s += a[k][e];
        if (s == 12345.678f) out[threadIdx.x] = s;
    } else if (MODE == 5) {
        f4v r0 = {0, 0, 0, 0};
        float acc = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(addr));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc += r0.x;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (acc == 12345.678f) out[threadIdx.x] = acc;
    } else if (MODE == 9) {
        float r = 0, acc = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                asm volatile("ds_max_f32 %1, %2\n\tds_min_rtn_f32 %0, %1, %3" : "=v"(r) : "v"(addr), "v"(acc + 1.0f), "v"(0.0f));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += r * 1e-30f;
        }
        if (acc == 12345.678f) out[threadIdx.x] = acc;
    } else if (MODE == 6) {
        float acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = 1.0f + k;
        for (int i = 0; i < iters * 24; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[k]) : "v"(1.0000001f));
        }
        float s = 0;
        for (int k = 0; k < 8; ++k) s += acc[k];
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
}

template <int MODE>
static void run(const char* what, int iters, float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mix<MODE>, dim3(256), dim3(512), 0, 0, iters, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix<MODE>, dim3(256), dim3(512), 0, 0, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 24;
    printf("%-78s %8.3f ms  %6.2f cycles/MFMA at 2.4 GHz  (%5.1f TFLOP/s)\n", what, ms, ms * 1e-3 * 2.4e9 / mfmas,
           256.0 * 4 * mfmas * 4096 / ms * 1e-9);
}

int main() {
    float* out; (void)hipMalloc(&out, 4096 * 4);
    const int iters = 4000;
    run<0>("MFMA only, 1 wave/SIMD, 8 accumulators", iters, out);
    run<1>("+ 1 ds_read_b128 per 3 MFMAs, same wave, lane*16 addresses", iters, out);
    run<2>("+ 2 ds_read_b64 per 3 MFMAs, same wave", iters, out);
    run<3>("+ 1 ds_read_b32 per 3 MFMAs, same wave", iters, out);
    run<4>("+ 1 ds_read_b128 per 3 MFMAs, same wave, one address per half-wave (broadcast)", iters, out);
    run<7>("+ 1 ds_read_b128 per MFMA, same wave", iters, out);
    run<8>("+ 1 ds_max_f32 + 1 ds_min_rtn_f32 per MFMA, same wave", iters, out);
    run<9>("MFMA wave + a second wave issuing ds_max_f32 + ds_min_rtn_f32 pairs (1 pair per MFMA)", iters, out);
    run<5>("MFMA wave + a second wave on the SIMD reading ds_read_b128 in a loop", iters, out);
    run<6>("MFMA wave + a second wave on the SIMD running v_mul_f32 back to back", iters, out);
    return 0;
}
