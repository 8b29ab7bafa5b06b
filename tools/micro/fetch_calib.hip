// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report on gfx950 for the access shapes of the SpMM kernel?
// MI355X_MICROARCH.md calibrates one shape only (16 B per lane streaming reads: FETCH_SIZE = bytes / 2) and calls the
// others uncalibrated.  Four kernels, each reading a known byte count ONCE from a buffer far larger than the caches'
// reach per launch (fresh region per launch):
//   k_read16: 16 B per lane, coalesced  (the calibrated shape)
//   k_read4 :  4 B per lane, coalesced  (column-index reads)
//   k_rows128: random 128-byte row pieces, 8 lanes x 16 B each (the SpMM gather)
//   k_rows128_seq: the same pieces in ascending row order
// Run: rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib ; compare the per-kernel counter with the byte count printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_read16(const float4* p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = i < n ? p[i] : make_float4(0, 0, 0, 0);
    if (v.x == 123.456f) sink[0] = v.y;
}
__global__ void k_read4(const int* p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int v = i < n ? p[i] : 0;
    if (v == 123456789) sink[0] = 1.0f;
}
__global__ void k_rows128(const float4* p, const int* rows, size_t npieces, float* sink) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t piece = t >> 3;
    if (piece >= npieces) return;
    float4 v = p[(size_t)rows[piece] * 8 + (t & 7)];
    if (v.x == 123.456f) sink[0] = v.y;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const size_t MB = 1 << 20, bytes = 512 * MB;        // each kernel touches its own 512 MB (> L2s + Infinity Cache)
    char* buf; float* sink; int* rows;
    CK(hipMalloc(&buf, 4 * bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, 4 * bytes));
    const size_t npieces = bytes / 128;
    std::vector<int> h(npieces);
    for (size_t i = 0; i < npieces; ++i) h[i] = (int)i;
    CK(hipMalloc(&rows, 2 * npieces * sizeof(int)));
    CK(hipMemcpy(rows + npieces, h.data(), npieces * sizeof(int), hipMemcpyHostToDevice));      // ascending
    srand(1);
    for (size_t i = npieces - 1; i > 0; --i) { size_t j = (size_t)rand() % (i + 1); int t = h[i]; h[i] = h[j]; h[j] = t; }
    CK(hipMemcpy(rows, h.data(), npieces * sizeof(int), hipMemcpyHostToDevice));                // a permutation: each piece once
    CK(hipDeviceSynchronize());
    const size_t n16 = bytes / 16, n4 = bytes / 4;
    hipLaunchKernelGGL(k_read16, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (const float4*)buf, n16, sink);
    hipLaunchKernelGGL(k_read4, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const int*)(buf + bytes), n4, sink);
    hipLaunchKernelGGL(k_rows128, dim3((unsigned)(npieces * 8 / 256)), dim3(256), 0, 0, (const float4*)(buf + 2 * bytes), rows, npieces, sink);
    hipLaunchKernelGGL(k_rows128, dim3((unsigned)(npieces * 8 / 256)), dim3(256), 0, 0, (const float4*)(buf + 3 * bytes), rows + npieces, npieces, sink);
    CK(hipDeviceSynchronize());
    printf("each kernel reads %zu bytes of payload (k_rows128 x2: + %zu bytes of row indices); order: read16, read4, rows128 random, rows128 ascending\n",
           bytes, npieces * sizeof(int));
    return 0;
}
