// How does a flat streaming kernel scale with the number of arrays it streams?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { const f4* r[24]; f4* w[12]; };
template<int R, int W> __global__ __launch_bounds__(256) void multi(Ptrs p, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n4) return;
    f4 v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = __builtin_nontemporal_load(p.r[k] + i);
    f4 s = {0,0,0,0};
#pragma unroll
    for (int k = 0; k < R; ++k) s += v[k];
#pragma unroll
    for (int k = 0; k < W; ++k) __builtin_nontemporal_store(s * (float)(k + 1), p.w[k] + i);
}
template<int R, int W> void run(Ptrs p, size_t n4, const char* tag) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((multi<R, W>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, p, n4);
    hipEventRecord(e0); for (int i = 0; i < 5; i++) hipLaunchKernelGGL((multi<R, W>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, p, n4);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%s R=%d W=%d: %.3f ms, %.0f GB/s\n", tag, R, W, ms, (double)(R + W) * n4 * 16 / ms / 1e6); fflush(stdout);
}
int main() {
    const size_t n = (size_t)512 * 512 * 512, n4 = n / 4;
    Ptrs p; 
    for (int k = 0; k < 24; k++) { void* q; hipMalloc(&q, n * 4); hipMemset(q, 0, n * 4); p.r[k] = (const f4*)q; }
    for (int k = 0; k < 12; k++) { void* q; hipMalloc(&q, n * 4); p.w[k] = (f4*)q; }
    run<6, 4>(p, n4, "2-D like"); run<15, 9>(p, n4, "3-D like"); run<8, 4>(p, n4, "mid"); run<12, 6>(p, n4, "mid2"); run<15, 9>(p, n4, "3-D like");
    // in-place variant: writes go back into read arrays
    Ptrs q = p; for (int k = 0; k < 9; k++) q.w[k] = (f4*)p.r[k];
    run<15, 9>(q, n4, "3-D like, in place");
    return 0;
}
