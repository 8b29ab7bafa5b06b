// Checks that v_pk_mul_f32 ... clamp computes clamp(x * s, 0, 1) the way the CPU does (IEEE multiply, denormals kept), so that
// relu(x) * w == clamp01(x * 2^-63) * (w * 2^63) bit for bit inside the exactness window.
// build: hipcc -O2 --offload-arch=gfx950 -o pk_clamp_relu tools/micro/pk_clamp_relu.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, const float* w, float* r, float* prod, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    f2v v = {x[2 * i], x[2 * i + 1]}, s = {0x1p-63f, 0x1p-63f}, o;
    asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(o) : "v"(v), "v"(s));
    r[2 * i] = o[0]; r[2 * i + 1] = o[1];
    prod[2 * i] = __builtin_fmaf(o[0], w[2 * i] * 0x1p63f, 0.25f);
    prod[2 * i + 1] = __builtin_fmaf(o[1], w[2 * i + 1] * 0x1p63f, 0.25f);
}
int main() {
    std::vector<float> x = {-1.0f, 0.0f, -0.0f, 1.0f, 3.5f, 1e-10f, 1e-19f, 1e-20f, 1e-30f, 1e-38f, 1e18f, 9e18f, 1e19f, 1e30f, NAN, -INFINITY,
                            INFINITY, 123.456f, 7.0e-5f, 2.5e-12f};
    for (int i = 0; i < 2000; ++i) x.push_back(ldexpf((float)(rand() % 20001 - 10000) / 10000.0f, rand() % 160 - 100));
    if (x.size() & 1) x.push_back(1.0f);
    const int n = (int)x.size();
    std::vector<float> w(n), r(n), p(n);
    for (int i = 0; i < n; ++i) w[i] = (float)(rand() % 2001 - 1000) / 777.0f;
    float *dx, *dw, *dr, *dp;
    hipMalloc(&dx, n * 4); hipMalloc(&dw, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dp, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n / 2 + 63) / 64), dim3(64), 0, 0, dx, dw, dr, dp, n);
    hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost); hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost);
    int bad_formula = 0, bad_relu = 0, in_window = 0;
    for (int i = 0; i < n; ++i) {
        float c = x[i] * 0x1p-63f;
        c = !(c > 0.0f) ? 0.0f : (c > 1.0f ? 1.0f : c);           // NaN -> 0 like DX10 clamp
        if (memcmp(&c, &r[i], 4) != 0 && !(c == 0.0f && r[i] == 0.0f)) {
            if (bad_formula++ < 10) printf("formula differs: x=%a gpu=%a cpu=%a\n", x[i], r[i], c);
        }
        const bool win = !(x[i] > 0.0f) || (x[i] >= 0x1p-63f && x[i] <= 0x1p63f);
        if (win && x[i] == x[i]) {
            ++in_window;
            const float relu = x[i] > 0.0f ? x[i] : 0.0f;
            const float want = fmaf(relu, w[i], 0.25f);
            if (memcmp(&want, &p[i], 4) != 0) { if (bad_relu++ < 10) printf("relu product differs: x=%a w=%a gpu=%a cpu=%a\n", x[i], w[i], p[i], want); }
        }
    }
    printf("%d values: clamp formula mismatches %d; in-window fma(relu(x), w, c) mismatches %d of %d\n", n, bad_formula, bad_relu, in_window);
    return bad_formula || bad_relu;
}
