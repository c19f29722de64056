// Does ALU work between stores overlap with the (HBM-bound) stores at k_expand's occupancy?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
constexpr int COLS = 250;
template <int K, bool STORE>
__global__ __launch_bounds__(256) void k_st(u64* out, size_t ld) {
    extern __shared__ char lds[];
    size_t sig = (size_t)blockIdx.x * 256 + threadIdx.x;
    u64* p = out + (size_t)(blockIdx.y * COLS) * ld + sig;
    u64 v = sig * 0x9E3779B97F4A7C15ull + blockIdx.y;
    unsigned x = (unsigned)v | 1, y = (unsigned)(v >> 32) | 1;
#pragma unroll 5
    for (int c = 0; c < COLS; c++) {
#pragma unroll
        for (int k = 0; k < K; k++) { v = (u64)x * y + v; x += (unsigned)(v >> 32); }
        if (STORE) *p = v; else if (v == 0xDEADBEEFull) *p = v;
        p += ld;
    }
    if (threadIdx.x == 9999) lds[0] = 1;
}
template <class F> float timeit(F f) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); f(); f(); f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / 3;
}
template <int K> void run(u64* d, size_t ld, size_t n, int segs, int ldsbytes) {
    float a = timeit([&] { hipLaunchKernelGGL((k_st<K, true>), dim3(n / 256, segs), dim3(256), ldsbytes, 0, d, ld); });
    float b = timeit([&] { hipLaunchKernelGGL((k_st<K, false>), dim3(n / 256, segs), dim3(256), ldsbytes, 0, d, ld); });
    printf("  K=%2d mads/store: with stores %.3f ms, ALU only %.3f ms\n", K, a, b);
}
int main() {
    const size_t n = 65536; const int segs = 328; size_t ld = n + 16;
    u64* d; (void)hipMalloc(&d, (size_t)segs * COLS * ld * 8);
    for (int ldsbytes : {0, 52000, 80000}) {   // 8 / 3 / 2 workgroups per CU -> 8 / 3 / 2 waves per SIMD
        (void)hipFuncSetAttribute((const void*)k_st<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
        printf("dynamic LDS %d B per workgroup:\n", ldsbytes);
        run<0>(d, ld, n, segs, ldsbytes); run<2>(d, ld, n, segs, ldsbytes); run<4>(d, ld, n, segs, ldsbytes);
        run<8>(d, ld, n, segs, ldsbytes); run<12>(d, ld, n, segs, ldsbytes);
    }
    return 0;
}
