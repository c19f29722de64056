// micro-benchmarks: cost of fe_mul and of raw v_mad_u64_u32 on gfx950, at 1 wave/SIMD and at high occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../plonky2-ecdsa_amd/csrc/ec.hpp"
using namespace p2e;

__global__ void k_mulchain(U256* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    U256 a = io[i], b = a;
    for (int k = 0; k < iters; k++) a = fe_mul<ModP>(a, b);
    io[i] = a;
}
__global__ void k_mulchain2(U256* io, int iters) {  // two independent chains per lane
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    U256 a = io[i], b = a, c = a;
    c.w[0] ^= 5;
    for (int k = 0; k < iters; k++) { a = fe_mul<ModP>(a, b); c = fe_mul<ModP>(c, b); }
    for (int k = 0; k < 8; k++) a.w[k] ^= c.w[k];
    io[i] = a;
}
__global__ void k_dbl(U256* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Jac p; p.X = io[i]; p.Y = io[i]; p.Y.w[0] ^= 3; p.Z = u256_small(1);
    for (int k = 0; k < iters; k++) p = jac_dbl(p).p;
    io[i] = p.X;
}
__global__ void k_mad(unsigned long long* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long a0 = io[i], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned x = (unsigned)a0 | 1, y = (unsigned)(a0 >> 32) | 1;
    for (int k = 0; k < iters; k++) {
        a0 = (unsigned long long)x * y + a0; a1 = (unsigned long long)x * y + a1;
        a2 = (unsigned long long)x * y + a2; a3 = (unsigned long long)x * y + a3;
        x += (unsigned)a0; y ^= (unsigned)a1;
    }
    io[i] = a0 ^ a1 ^ a2 ^ a3;
}
__global__ void k_add(unsigned* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned a0 = io[i], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int k = 0; k < iters; k++) { a0 = a0 * 3 + a1; a1 = a1 ^ (a2 >> 3); a2 = a2 + a3; a3 = a3 - (a0 << 2); }
    io[i] = a0 ^ a1 ^ a2 ^ a3;
}
template <class F> float timeit(F f) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}
int main() {
    const int maxthreads = 256 * 4 * 8 * 64;  // 8 waves/SIMD
    U256* d; hipMalloc(&d, sizeof(U256) * maxthreads);
    std::vector<U256> h(maxthreads);
    for (int i = 0; i < maxthreads; i++) for (int k = 0; k < 8; k++) h[i].w[k] = 0x9E3779B9u * (i * 8 + k + 1);
    hipMemcpy(d, h.data(), sizeof(U256) * maxthreads, hipMemcpyHostToDevice);
    const double clk = 2.4e9;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;  // 256 CUs x wps blocks of 256 threads = wps waves per SIMD
        int iters = 2000;
        float ms = timeit([&] { hipLaunchKernelGGL(k_mulchain, dim3(blocks), dim3(256), 0, 0, d, iters); });
        printf("fe_mul chain   waves/SIMD=%d: %.3f ms  -> %.0f cycles per fe_mul per wave, %.1f cycles/SIMD-slot\n", wps, ms, ms * 1e-3 * clk / iters, ms * 1e-3 * clk / iters / wps);
        ms = timeit([&] { hipLaunchKernelGGL(k_mulchain2, dim3(blocks), dim3(256), 0, 0, d, iters); });
        printf("fe_mul 2chains waves/SIMD=%d: %.3f ms  -> %.0f cycles per fe_mul per wave\n", wps, ms, ms * 1e-3 * clk / iters / 2);
        ms = timeit([&] { hipLaunchKernelGGL(k_dbl, dim3(blocks), dim3(256), 0, 0, d, 400); });
        printf("jac_dbl chain  waves/SIMD=%d: %.3f ms  -> %.0f cycles per dbl per wave\n", wps, ms, ms * 1e-3 * clk / 400);
        ms = timeit([&] { hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, (unsigned long long*)d, 20000); });
        printf("mad_u64_u32    waves/SIMD=%d: %.3f ms  -> %.2f cycles per mad per wave, %.2f per SIMD-slot\n", wps, ms, ms * 1e-3 * clk / (20000 * 4), ms * 1e-3 * clk / (20000 * 4) / wps);
        ms = timeit([&] { hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, 0, (unsigned*)d, 20000); });
        printf("int alu (6 ops) waves/SIMD=%d: %.3f ms -> %.2f cycles per iter per wave\n", wps, ms, ms * 1e-3 * clk / 20000);
    }
    return 0;
}
