// Does the size of the workgroup (= contiguous bytes per column written by one workgroup pass) change what the
// k_expand store pattern can sustain?  Paired 16-byte stores as in PairEmit, COLS columns per workgroup row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
constexpr int COLS = 250;
template <int BSZ>
__global__ __launch_bounds__(BSZ) void k_pair(u64* out, size_t ld) {
    unsigned t = threadIdx.x;
    size_t sig = (size_t)blockIdx.x * BSZ + (t & ~63u) + 2u * (t & 31u) + ((t >> 5) & 1u);
    unsigned upper = sig & 1;
    u64* base = out + (size_t)(blockIdx.y * COLS) * ld;
    u64 v = sig * 0x9E3779B97F4A7C15ull + blockIdx.y;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#pragma unroll 10
    for (int c = 0; c < COLS; c += 2) {
        v4u o = {(unsigned)v, (unsigned)(v >> 32), (unsigned)v + 1, 7u};
        __builtin_nontemporal_store(o, reinterpret_cast<v4u*>(base + (size_t)(c + upper) * ld + (sig - upper)));
        v += 0x1234567;
    }
}
template <class F> float timeit(F f) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); f(); f(); f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / 3;
}
int main() {
    const size_t n = 65536; const int segs = 328;
    size_t ld = n + 16;
    u64* d; (void)hipMalloc(&d, (size_t)segs * COLS * ld * 8);
    double bytes = (double)segs * COLS * n * 8;
    float a = timeit([&] { hipLaunchKernelGGL(k_pair<256>, dim3(n / 256, segs), dim3(256), 0, 0, d, ld); });
    float b = timeit([&] { hipLaunchKernelGGL(k_pair<512>, dim3(n / 512, segs), dim3(512), 0, 0, d, ld); });
    float c = timeit([&] { hipLaunchKernelGGL(k_pair<1024>, dim3(n / 1024, segs), dim3(1024), 0, 0, d, ld); });
    float e = timeit([&] { hipLaunchKernelGGL(k_pair<128>, dim3(n / 128, segs), dim3(128), 0, 0, d, ld); });
    float f = timeit([&] { hipLaunchKernelGGL(k_pair<64>, dim3(n / 64, segs), dim3(64), 0, 0, d, ld); });
    printf("paired nt stores, workgroup 64: %.2f TB/s | 128: %.2f | 256: %.2f | 512: %.2f | 1024: %.2f\n", bytes / f / 1e9, bytes / e / 1e9,
           bytes / a / 1e9, bytes / b / 1e9, bytes / c / 1e9);
    (void)hipFree(d);
    return 0;
}
