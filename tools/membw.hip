// HBM bandwidth microbenchmarks: what can a streaming kernel reach on this box?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef float f4 __attribute__((ext_vector_type(4)));

template<int NT> __global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
__global__ __launch_bounds__(256) void read_k(const f4* __restrict__ a, float* out, size_t n4) {
    size_t stride = (size_t)gridDim.x * 256; f4 acc = {0,0,0,0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1;
}
__global__ __launch_bounds__(256) void write_k(f4* __restrict__ b, size_t n4) {
    size_t stride = (size_t)gridDim.x * 256; f4 v = {1,2,3,4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) b[i] = v;
}
// 7 read streams + 4 write streams, flat (one thread per float4, no stride loop)
template<int NT> __global__ __launch_bounds__(256) void multi_k(const f4* __restrict__ r0, const f4* __restrict__ r1, const f4* __restrict__ r2,
    const f4* __restrict__ r3, const f4* __restrict__ r4, const f4* __restrict__ r5, f4* __restrict__ w0, f4* __restrict__ w1, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n4) return;
    f4 a, b, c, d, e, f;
    if (NT) { a=__builtin_nontemporal_load(r0+i); b=__builtin_nontemporal_load(r1+i); c=__builtin_nontemporal_load(r2+i); d=__builtin_nontemporal_load(r3+i); e=__builtin_nontemporal_load(r4+i); f=__builtin_nontemporal_load(r5+i);} 
    else { a=r0[i]; b=r1[i]; c=r2[i]; d=r3[i]; e=r4[i]; f=r5[i]; }
    f4 s = a + b + c, t = d + e + f;
    // in place for two of them (like ez_w, dz_w), out of place for two (hx_w, hy_w)
    if (NT) { __builtin_nontemporal_store(s, w0+i); __builtin_nontemporal_store(t, w1+i); __builtin_nontemporal_store(s+t, (f4*)r3+i); __builtin_nontemporal_store(s-t, (f4*)r5+i);} 
    else { w0[i] = s; w1[i] = t; ((f4*)r3)[i] = s + t; ((f4*)r5)[i] = s - t; }
}
int main() {
    const size_t n = (size_t)16384 * 16384, n4 = n / 4, bytes = n * 4;
    float* p[8]; for (auto& q : p) { CK(hipMalloc(&q, bytes)); CK(hipMemset(q, 0, bytes)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double gb, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0); for (int i = 0; i < 10; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-40s %8.3f ms  %7.0f GB/s\n", name, ms, gb / ms * 1e3 / 1e9 * 1e-3 * 1e3); fflush(stdout);
    };
    for (int blocks : {1024, 2048, 4096, 8192, 65536}) {
        char nm[64]; snprintf(nm, 64, "copy f4 grid-stride %d blocks", blocks);
        timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_k<0>, dim3(blocks), dim3(256), 0, 0, (const f4*)p[0], (f4*)p[1], n4); });
    }
    timeit("copy f4 nontemporal 4096 blocks", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_k<1>, dim3(4096), dim3(256), 0, 0, (const f4*)p[0], (f4*)p[1], n4); });
    timeit("copy f4 flat (1 thread per f4)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_k<0>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const f4*)p[0], (f4*)p[1], n4); });
    timeit("read f4 4096 blocks", 1.0 * bytes, [&] { hipLaunchKernelGGL(read_k, dim3(4096), dim3(256), 0, 0, (const f4*)p[0], p[7], n4); });
    timeit("write f4 4096 blocks", 1.0 * bytes, [&] { hipLaunchKernelGGL(write_k, dim3(4096), dim3(256), 0, 0, (f4*)p[1], n4); });
    timeit("hipMemcpyDtoD", 2.0 * bytes, [&] { hipMemcpyAsync(p[1], p[0], bytes, hipMemcpyDeviceToDevice, 0); });
    timeit("multi 6R+4W flat", 10.0 * bytes, [&] { hipLaunchKernelGGL(multi_k<0>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const f4*)p[0], (const f4*)p[1], (const f4*)p[2], (const f4*)p[3], (const f4*)p[4], (const f4*)p[5], (f4*)p[6], (f4*)p[7], n4); });
    timeit("multi 6R+4W flat nontemporal", 10.0 * bytes, [&] { hipLaunchKernelGGL(multi_k<1>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const f4*)p[0], (const f4*)p[1], (const f4*)p[2], (const f4*)p[3], (const f4*)p[4], (const f4*)p[5], (f4*)p[6], (f4*)p[7], n4); });
    return 0;
}
