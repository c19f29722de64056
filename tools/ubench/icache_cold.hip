// What does a kernel pay for executing straight-line code ONCE?  Every launch starts with a cold instruction cache; a
// lone wave then waits for each 64-byte line of code it has never seen.  Two kernels do the same 48 dependent lazy-limb
// multiplications (fe29.hpp, ~1.4 KB of code each): one as a rolled loop (the body is fetched once), one fully unrolled
// (~70 KB of code).  The difference per multiplication is the price of fetching 1.4 KB of cold code.
// Also: the same with one wave per SIMD on the whole chip (every CU pair fetches its own copy).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/icache_cold tools/ubench/icache_cold.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../plonky2-ecdsa_amd/csrc/fe29.hpp"
using namespace p2e;
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
constexpr int N = 48;
__global__ void k_rolled(const U256* in, U256* out, unsigned long long* ticks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    F29 x = f29_from_u256(in[2 * i]), y = f29_from_u256(in[2 * i + 1]);
    const unsigned long long t0 = wall_clock64();
#pragma nounroll
    for (int k = 0; k < N; k++) x = f29_mul(x, y);
    const unsigned long long t1 = wall_clock64();
    out[i] = f29_canon(x);
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int K>
__device__ __forceinline__ void chain(F29& x, const F29& y) {
    x = f29_mul(x, y);
    if constexpr (K > 1) chain<K - 1>(x, y);
}
__global__ void k_unrolled(const U256* in, U256* out, unsigned long long* ticks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    F29 x = f29_from_u256(in[2 * i]), y = f29_from_u256(in[2 * i + 1]);
    const unsigned long long t0 = wall_clock64();
    chain<N>(x, y);
    const unsigned long long t1 = wall_clock64();
    out[i] = f29_canon(x);
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    const int max_blocks = 1024;
    std::vector<U256> in((size_t)max_blocks * 64 * 2);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : in) {
        for (int k = 0; k < 8; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v.w[k] = (u32)(s >> 16); }
        v.w[7] &= 0x7FFFFFFFu;
    }
    U256 *d_in, *d_out;
    unsigned long long* d_ticks;
    CK(hipMalloc(&d_in, in.size() * 32));
    CK(hipMalloc(&d_out, (size_t)max_blocks * 64 * 32));
    CK(hipMalloc(&d_ticks, max_blocks * 8));
    CK(hipMemcpy(d_in, in.data(), in.size() * 32, hipMemcpyHostToDevice));
    std::vector<unsigned long long> ticks(max_blocks);
    for (int blocks : {1, 1024}) {
        for (int which = 0; which < 2; which++) {
            double best = 1e30, worst = 0;
            for (int rep = 0; rep < 5; rep++) {
                if (which == 0) hipLaunchKernelGGL(k_rolled, dim3(blocks), dim3(64), 0, 0, d_in, d_out, d_ticks);
                else hipLaunchKernelGGL(k_unrolled, dim3(blocks), dim3(64), 0, 0, d_in, d_out, d_ticks);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(ticks.data(), d_ticks, blocks * 8, hipMemcpyDeviceToHost));
                unsigned long long w = 0;
                for (int b = 0; b < blocks; b++) w = ticks[b] > w ? ticks[b] : w;
                const double ns = w * 10.0 / N;
                best = ns < best ? ns : best;
                worst = ns > worst ? ns : worst;
            }
            printf("%-34s %-10s %8.1f ... %8.1f ns per multiplication (slowest wave; best and worst of 5 launches)\n",
                   blocks == 1 ? "one wave on the chip" : "one wave on every SIMD", which == 0 ? "rolled" : "unrolled", best, worst);
        }
    }
    return 0;
}
