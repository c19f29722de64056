// How fast can the k_expand store pattern go with no ALU work?  out[col * ld + sig], 65536 sigs,
// every workgroup (256 lanes = 256 sigs) writes COLS consecutive columns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
constexpr int COLS = 250;

__global__ __launch_bounds__(256) void k_st8(u64* out, size_t ld, int seg) {
    size_t sig = (size_t)blockIdx.x * 256 + threadIdx.x;
    u64* p = out + (size_t)(blockIdx.y * COLS) * ld + sig;
    u64 v = sig * 0x9E3779B97F4A7C15ull + blockIdx.y;
#pragma unroll 10
    for (int c = 0; c < COLS; c++) { *p = v; p += ld; v += 0x1234567; }
}
__global__ __launch_bounds__(256) void k_st16pair(u64* out, size_t ld, int seg) {   // PairEmit pattern
    unsigned t = threadIdx.x;
    size_t sig = (size_t)blockIdx.x * 256 + (t & ~63u) + 2u * (t & 31u) + ((t >> 5) & 1u);
    unsigned upper = sig & 1;
    u64* base = out + (size_t)(blockIdx.y * COLS) * ld;
    u64 v = sig * 0x9E3779B97F4A7C15ull + blockIdx.y;
#pragma unroll 10
    for (int c = 0; c < COLS; c += 2) {
        uint4 o = make_uint4((unsigned)v, (unsigned)(v >> 32), (unsigned)v + 1, 7u);
        *reinterpret_cast<uint4*>(base + (size_t)(c + upper) * ld + (sig - upper)) = o;
        v += 0x1234567;
    }
}
__global__ __launch_bounds__(256) void k_st16x2(u64* out, size_t ld, int seg) {   // each lane: 2 adjacent sigs, one column per store
    size_t sig2 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    u64* p = out + (size_t)(blockIdx.y * COLS) * ld + sig2;
    u64 v = sig2 * 0x9E3779B97F4A7C15ull + blockIdx.y;
#pragma unroll 10
    for (int c = 0; c < COLS; c++) {
        *reinterpret_cast<uint4*>(p) = make_uint4((unsigned)v, (unsigned)(v >> 32), (unsigned)v + 1, 7u);
        p += ld; v += 0x1234567;
    }
}
// reference points: a perfectly linear fill (every wave writes 1 KB, waves walk the buffer in order), with plain and
// with non-temporal stores
template <bool NT>
__global__ __launch_bounds__(256) void k_linear(uint4* out, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    v4u v = {(unsigned)i, 1u, 2u, 3u};
    for (; i < n16; i += step) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(out + i));
        else *reinterpret_cast<v4u*>(out + i) = v;
    }
}
template <class F> float timeit(F f) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); f(); f(); f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / 3;
}
int main() {
    const size_t n = 65536; const int segs = 328;   // 328 * 250 = 82000 columns
    for (size_t pad : {(size_t)0, (size_t)16}) {
        size_t ld = n + pad;
        u64* d; (void)hipMalloc(&d, (size_t)segs * COLS * ld * 8);
        double bytes = (double)segs * COLS * n * 8;
        float a = timeit([&] { hipLaunchKernelGGL(k_st8, dim3(n / 256, segs), dim3(256), 0, 0, d, ld, 0); });
        float b = timeit([&] { hipLaunchKernelGGL(k_st16pair, dim3(n / 256, segs), dim3(256), 0, 0, d, ld, 0); });
        float c = timeit([&] { hipLaunchKernelGGL(k_st16x2, dim3(n / 512, segs), dim3(256), 0, 0, d, ld, 0); });
        printf("pad=%zu: 8B/lane %.3f ms %.2f TB/s | 16B paired-columns %.3f ms %.2f TB/s | 16B 2-sigs/lane %.3f ms %.2f TB/s\n", pad,
               a, bytes / a / 1e9, b, bytes / b / 1e9, c, bytes / c / 1e9);
        size_t n16 = (size_t)segs * COLS * ld / 2;
        for (unsigned blocks : {2048u, 8192u, 65536u}) {
            float l0 = timeit([&] { hipLaunchKernelGGL(k_linear<false>, dim3(blocks), dim3(256), 0, 0, (uint4*)d, n16); });
            float l1 = timeit([&] { hipLaunchKernelGGL(k_linear<true>, dim3(blocks), dim3(256), 0, 0, (uint4*)d, n16); });
            printf("   linear fill, %u blocks: plain %.3f ms %.2f TB/s | nt %.3f ms %.2f TB/s\n", blocks, l0, n16 * 16.0 / l0 / 1e9, l1,
                   n16 * 16.0 / l1 / 1e9);
        }
        float m = timeit([&] { (void)hipMemsetAsync(d, 0x5a, n16 * 16, 0); });
        printf("   hipMemsetAsync %.3f ms %.2f TB/s\n", m, n16 * 16.0 / m / 1e9);
        (void)hipFree(d);
    }
    return 0;
}
