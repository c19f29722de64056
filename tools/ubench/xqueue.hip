// Dependency latency between kernels: how long after kernel A has ended does a dependent kernel B start, when B sits
//   (0) on the same stream,
//   (1) on another stream behind hipStreamWaitEvent on an event with timing disabled,
//   (2) the same with a timing-enabled event,
//   (3) as (1) with B's stream created at high priority, (4) as (1) with a third stream running an unrelated long kernel.
// Times from the device's constant 100 MHz counter (wall_clock64) written by the kernels themselves.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/xqueue tools/ubench/xqueue.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ void k_busy(unsigned long long* t, int slot, int spin) {
    if (threadIdx.x == 0 && blockIdx.x == 0) t[2 * slot] = wall_clock64();
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) {}
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) t[2 * slot + 1] = wall_clock64();
}
int main() {
    unsigned long long* t;
    CK(hipHostMalloc(&t, 4096 * sizeof(unsigned long long)));
    hipStream_t s1, s2, s2hi, s3;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s2hi, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    hipEvent_t e_nt, e_t;
    CK(hipEventCreateWithFlags(&e_nt, hipEventDisableTiming));
    CK(hipEventCreate(&e_t));
    const int spinA = 20000 /* 200 us */, reps = 40;
    const char* names[5] = {"same stream", "other stream, event without timing", "other stream, timing event", "other stream (high priority)", "other stream, third stream busy"};
    for (int mode = 0; mode < 5; mode++) {
        std::vector<double> gaps;
        for (int r = 0; r < reps + 3; r++) {
            hipStream_t sb = mode == 0 ? s1 : mode == 3 ? s2hi : s2;
            if (mode == 4) hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, s3, t, 2, 100000);
            hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s1, t, 0, spinA);
            if (mode != 0) {
                hipEvent_t e = mode == 2 ? e_t : e_nt;
                CK(hipEventRecord(e, s1));
                CK(hipStreamWaitEvent(sb, e, 0));
            }
            hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, sb, t, 1, 1000);
            CK(hipDeviceSynchronize());
            if (r >= 3) gaps.push_back((double)(long long)(t[2] - t[1]) / 100.0);   // us
        }
        std::sort(gaps.begin(), gaps.end());
        printf("%-40s gap A.end -> B.start: median %7.1f us  min %7.1f  max %7.1f\n", names[mode], gaps[gaps.size() / 2], gaps.front(), gaps.back());
    }
    return 0;
}
