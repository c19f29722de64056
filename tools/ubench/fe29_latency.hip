// Latency of the field operations as a LONE WAVE sees them (the small-batch plan has at most one chain wave per SIMD):
// fe.hpp's canonical 8 x 32-bit form against fe29.hpp's lazy 9 x 29-bit form.  Each kernel runs a dependent chain of
// `iters` operations in one wave (grid 1 x 64) and the host reports ns per operation from the device's 100 MHz counter;
// the same with one wave on every SIMD (grid 1024 x 64) shows what changes when the chip's clock is under load.
// Results of both forms are compared at the end (canonical words), so a wrong variant cannot look fast.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fe29_latency tools/ubench/fe29_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../plonky2-ecdsa_amd/csrc/fe29.hpp"
using namespace p2e;
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

enum { OP_MUL, OP_SQR, OP_ADD, OP_SUB, OP_DBLISH, OP_CANON, OP_INV, OP_COUNT };
static const char* op_name[OP_COUNT] = {"mul", "sqr", "add", "sub", "doubling-like mix (3 mul/sqr + 14 add/sub)", "store form (canon / none)", "inversion (fe_inv: safegcd; both columns: words)"};

// canonical words
template <int OP>
__global__ void k_words(const U256* in, U256* out, unsigned long long* ticks, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    U256 x = in[2 * i], y = in[2 * i + 1];
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < iters; k++) {
        if (OP == OP_MUL) x = fe_mul<ModP>(x, y);
        if (OP == OP_SQR) x = fe_sqr<ModP>(x);
        if (OP == OP_ADD) x = fe_add<ModP>(x, y);
        if (OP == OP_SUB) x = fe_sub<ModP>(x, y);
        if (OP == OP_DBLISH) {   // the shape of jac_dbl_quad for one lane: 3 levels, 9 additions, 5 subtractions
            const U256 a = fe_mul<ModP>(x, y);
            const U256 z3 = fe_add<ModP>(a, a);
            const U256 e = fe_add<ModP>(fe_add<ModP>(a, a), a);
            const U256 f = fe_sqr<ModP>(fe_add<ModP>(x, e));
            const U256 t = fe_sub<ModP>(fe_sub<ModP>(f, a), y);
            const U256 d = fe_add<ModP>(t, t);
            const U256 x3 = fe_sub<ModP>(f, fe_add<ModP>(d, d));
            const U256 c2 = fe_add<ModP>(a, a), c4 = fe_add<ModP>(c2, c2), c8 = fe_add<ModP>(c4, c4);
            const U256 y0 = fe_mul<ModP>(e, fe_sub<ModP>(d, x3));
            x = fe_sub<ModP>(y0, c8);
            y = fe_add<ModP>(x3, z3);
        }
        if (OP == OP_CANON) x = u256_select(u256_is_zero(x), y, x);
        if (OP == OP_INV) x = fe_add<ModP>(fe_inv<ModP>(x), y);
    }
    const unsigned long long t1 = wall_clock64();
    out[i] = x;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
// lazy limbs
template <int OP>
__global__ void k_limbs(const U256* in, U256* out, unsigned long long* ticks, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    F29 x = f29_from_u256(in[2 * i]), y = f29_from_u256(in[2 * i + 1]);
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < iters; k++) {
        if (OP == OP_MUL) x = f29_mul(x, y);
        if (OP == OP_SQR) x = f29_sqr(x);
        if (OP == OP_ADD) x = f29_norm(f29_add(x, y));
        if (OP == OP_SUB) x = f29_norm(f29_sub<1>(x, y));
        if (OP == OP_DBLISH) {
            const F29 a = f29_mul(x, y);
            const F29 z3 = f29_times<2>(a);
            const F29 e = f29_norm(f29_times<3>(a));
            const F29 f = f29_sqr(f29_add(x, e));
            const F29 t = f29_norm(f29_sub<1>(f29_sub<1>(f, a), y));
            const F29 d = f29_times<2>(t);
            const F29 x3 = f29_norm(f29_sub<4>(f, f29_times<2>(d)));
            const F29 c8 = f29_times<2>(f29_norm(f29_times<4>(a)));
            const F29 y0 = f29_mul(e, f29_sub<1>(d, x3));
            x = f29_norm(f29_sub<2>(y0, c8));
            y = f29_norm(f29_add(x3, z3));
        }
        if (OP == OP_INV) x = f29_from_u256(fe_add<ModP>(fe_inv<ModP>(f29_canon(x)), f29_canon(y)));
        if (OP == OP_CANON) {   // what one op of the chain pays for its stores: one value to canonical words and back
            const U256 c = f29_canon(x);
            x = f29_select(u256_is_zero(c), y, f29_from_u256(c));
        }
    }
    const unsigned long long t1 = wall_clock64();
    out[i] = f29_canon(f29_norm(x));
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int OP>
static int run_op(const U256* d_in, U256* d_out, unsigned long long* d_ticks, int blocks, int iters_in, std::vector<U256>& res, double& ns_words,
                  double& ns_limbs) {
    const int iters = OP == OP_INV ? 40 : iters_in;
    std::vector<unsigned long long> ticks(blocks);
    std::vector<U256> a((size_t)blocks * 64), b((size_t)blocks * 64);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_words<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, d_ticks, iters);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(ticks.data(), d_ticks, blocks * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a.data(), d_out, a.size() * 32, hipMemcpyDeviceToHost));
    unsigned long long worst = 0;
    for (auto t : ticks) worst = t > worst ? t : worst;
    ns_words = worst * 10.0 / iters;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_limbs<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, d_ticks, iters);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(ticks.data(), d_ticks, blocks * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_out, b.size() * 32, hipMemcpyDeviceToHost));
    worst = 0;
    for (auto t : ticks) worst = t > worst ? t : worst;
    ns_limbs = worst * 10.0 / iters;
    if (OP != OP_CANON && std::memcmp(a.data(), b.data(), a.size() * 32) != 0) {
        printf("MISMATCH between the two forms in %s\n", op_name[OP]);
        return 2;
    }
    (void)res;
    return 0;
}

int main() {
    const int max_blocks = 1024, iters = 2000;
    std::vector<U256> in((size_t)max_blocks * 64 * 2);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : in) {
        for (int k = 0; k < 8; k++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            v.w[k] = (u32)(s >> 16);
        }
        v.w[7] &= 0x7FFFFFFFu;   // below p
    }
    U256 *d_in, *d_out;
    unsigned long long* d_ticks;
    CK(hipMalloc(&d_in, in.size() * 32));
    CK(hipMalloc(&d_out, (size_t)max_blocks * 64 * 32));
    CK(hipMalloc(&d_ticks, max_blocks * 8));
    CK(hipMemcpy(d_in, in.data(), in.size() * 32, hipMemcpyHostToDevice));
    std::vector<U256> res;
    for (int blocks : {1, 1024}) {
        printf("%s\n", blocks == 1 ? "one wave on the chip" : "one wave on every SIMD (1024 waves)");
        printf("  %-50s %12s %12s %8s\n", "operation", "words ns/op", "limbs ns/op", "ratio");
        double w, l;
        int rc = 0;
#define ROW(OP) rc = run_op<OP>(d_in, d_out, d_ticks, blocks, iters, res, w, l); if (rc) return rc; printf("  %-50s %12.1f %12.1f %8.2f\n", op_name[OP], w, l, l / w);
        ROW(OP_MUL) ROW(OP_SQR) ROW(OP_ADD) ROW(OP_SUB) ROW(OP_DBLISH) ROW(OP_CANON) ROW(OP_INV)
    }
    return 0;
}
