// Micro-benchmark: how many matrix-pipe cycles does one vector-ALU instruction cost a SIMD that is streaming f32 MFMAs
// (v_mfma_f32_32x32x2_f32, 64 cycles each) on gfx950?  (a) the VALU instructions come from the MFMA wave itself, placed between
// the MFMAs; (b) they come from a second wave on the same SIMD.  Reported: cycles per MFMA and the extra cycles per VALU op.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_cost tools/micro/mfma_valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

#define OPS(X) X(0, "none", "") X(1, "v_mul_f32", "v_mul_f32 %0, %0, %1") X(2, "v_mov_b32", "v_mov_b32 %0, %1") \
    X(3, "v_and_b32", "v_and_b32 %0, %0, %1") X(4, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1") X(5, "v_exp_f32", "v_exp_f32 %0, %1") \
    X(6, "v_add_u32", "v_add_u32 %0, %0, %1") X(7, "v_max_f32", "v_max_f32 %0, %0, %1")

template <int OP> __device__ __forceinline__ void vop(float& d, float s) {
#define X(id, name, text) if (OP == id && id != 0) asm volatile(text : "+v"(d) : "v"(s));
    OPS(X)
#undef X
}
__device__ __forceinline__ void vop_pk(f2v& d, f2v s, int which) {
    if (which == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d) : "v"(s));
    else asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(s));
}

// OP: VALU op; PER: ops per MFMA issued by the MFMA wave itself; PARTNER: ops per MFMA-time issued by a second wave (waves 4-7)
template <int OP, int PER, int PARTNER, int PK>
__global__ void __launch_bounds__(512)
k_cost(int iters, float* out) {
    const int wv = threadIdx.x >> 6;
    float t[8];
    f2v p[8];
    for (int k = 0; k < 8; ++k) { t[k] = 1.0f + k + threadIdx.x; p[k] = f2v{t[k], t[k] + 1}; }
    const float one = 1.0000001f;
    const f2v one2 = {one, one};
    if (wv < 4) {
        f16v a[8];
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 16; ++e) a[k][e] = 0.f;
        float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                a[k & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[k & 7], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < PER; ++r) {
                    if (PK) vop_pk(p[(k * PER + r) & 7], one2, PK - 1); else vop<OP>(t[(k * PER + r) & 7], one);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float s = 0;
        for (int k = 0; k < 8; ++k) { for (int e = 0; e < 16; ++e) s += a[k][e]; s += t[k] + p[k].x + p[k].y; }
        if (s == 12345.678f) out[threadIdx.x] = s;
    } else if (PARTNER > 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16 * PARTNER; ++k) {
                if (PK) vop_pk(p[k & 7], one2, PK - 1); else vop<OP>(t[k & 7], one);
            }
            __builtin_amdgcn_s_sleep(PARTNER >= 8 ? 1 : 8);
        }
        float s = 0;
        for (int k = 0; k < 8; ++k) s += t[k] + p[k].x + p[k].y;
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
}

static double base_cycles = 0;
template <int OP, int PER, int PARTNER, int PK>
static void run(const char* what, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_cost<OP, PER, PARTNER, PK>), dim3(256), dim3(512), 0, 0, iters, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_cost<OP, PER, PARTNER, PK>), dim3(256), dim3(512), 0, 0, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 16);
    if (PER == 0 && PARTNER == 0) base_cycles = cyc;
    const int nops = PER + PARTNER;
    printf("%-64s %8.3f ms  %7.2f cycles/MFMA", what, ms, cyc);
    if (nops) printf("  -> %+6.2f cycles per VALU op", (cyc - base_cycles) / nops);
    printf("\n");
}

int main() {
    float* out; (void)hipMalloc(&out, 4096 * 4);
    run<0, 0, 0, 0>("MFMA only (1 wave/SIMD)", out);
    run<1, 1, 0, 0>("same wave: 1 v_mul_f32 per MFMA", out);
    run<1, 4, 0, 0>("same wave: 4 v_mul_f32 per MFMA", out);
    run<1, 8, 0, 0>("same wave: 8 v_mul_f32 per MFMA", out);
    run<2, 4, 0, 0>("same wave: 4 v_mov_b32 per MFMA", out);
    run<3, 4, 0, 0>("same wave: 4 v_and_b32 per MFMA", out);
    run<6, 4, 0, 0>("same wave: 4 v_add_u32 per MFMA", out);
    run<7, 4, 0, 0>("same wave: 4 v_max_f32 per MFMA", out);
    run<4, 4, 0, 0>("same wave: 4 v_fma_f32 per MFMA", out);
    run<5, 2, 0, 0>("same wave: 2 v_exp_f32 per MFMA", out);
    run<0, 4, 0, 1>("same wave: 4 v_pk_mul_f32 per MFMA", out);
    run<0, 4, 0, 2>("same wave: 4 v_pk_fma_f32 per MFMA", out);
    run<1, 0, 1, 0>("partner wave: 1 v_mul_f32 per MFMA", out);
    run<1, 0, 4, 0>("partner wave: 4 v_mul_f32 per MFMA", out);
    run<1, 0, 8, 0>("partner wave: 8 v_mul_f32 per MFMA", out);
    run<2, 0, 4, 0>("partner wave: 4 v_mov_b32 per MFMA", out);
    run<3, 0, 4, 0>("partner wave: 4 v_and_b32 per MFMA", out);
    run<4, 0, 4, 0>("partner wave: 4 v_fma_f32 per MFMA", out);
    run<0, 0, 4, 2>("partner wave: 4 v_pk_fma_f32 per MFMA", out);
    return 0;
}
