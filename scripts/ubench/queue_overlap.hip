// Does a kernel with a huge grid on stream B hold up short dependent kernels on stream A?
// A: 200 x (120 blocks x 256 threads, ~8 us of s_sleep each), back to back.
// B: one kernel of ~600 us, either as 40000 short blocks or as 1024 persistent blocks doing the
//    same total work (VALU spin), or nothing.
// Reports the time stream A needs for its 200 kernels in each case.
//     hipcc --offload-arch=gfx950 -O3 queue_overlap.hip -o queue_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__device__ __forceinline__ unsigned long long rt()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__global__ void k_short(int *p)   // latency-bound stand-in: sleeps ~8 us
{
    const unsigned long long t0 = rt();
    while (rt() - t0 < 800) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 9999) *p = 1;
}
__global__ __launch_bounds__(256) void k_work(float *out, int items_per_block, int iters)   // VALU-bound
{
    float acc = threadIdx.x;
    for (int it = 0; it < items_per_block; ++it)
        for (int i = 0; i < iters; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 0.25f);
    if (acc == 12345.f) out[0] = acc;
}
int main()
{
    int *p; float *o;
    (void)hipMalloc(&p, 64); (void)hipMalloc(&o, 64);
    hipStream_t a, b;
    (void)hipStreamCreate(&a); (void)hipStreamCreate(&b);
    auto run_a = [&]() {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_short, 120, 256, 0, a, p);
        (void)hipStreamSynchronize(a);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    auto time_b = [&](int blocks, int items) {
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_work, blocks, 256, 0, b, o, items, 2000);
        (void)hipStreamSynchronize(b);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    run_a(); time_b(1024, 1);
    printf("A alone: %.0f us for 200 kernels\n", run_a());
    printf("B alone: 40960 blocks x 1 item: %.0f us;  1024 blocks x 40 items: %.0f us\n", time_b(40960, 1), time_b(1024, 40));
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipDeviceSynchronize();
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_work, 40960, 256, 0, b, o, 1, 2000);
        double ta = run_a();
        (void)hipDeviceSynchronize();
        printf("A while B runs 4 x (40960 short blocks): %.0f us\n", ta);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_work, 1024, 256, 0, b, o, 40, 2000);
        ta = run_a();
        (void)hipDeviceSynchronize();
        printf("A while B runs 4 x (1024 persistent blocks): %.0f us\n", ta);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_work, 512, 256, 0, b, o, 80, 2000);
        ta = run_a();
        (void)hipDeviceSynchronize();
        printf("A while B runs 4 x (512 persistent blocks): %.0f us\n", ta);
    }
    // does the choice of stream (hardware queue / pipe) or its priority matter?
    hipStream_t extra[8];
    for (int i = 0; i < 8; ++i) (void)hipStreamCreate(&extra[i]);
    for (int i = 0; i < 8; ++i) {
        (void)hipDeviceSynchronize();
        for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(k_work, 40960, 256, 0, extra[i], o, 1, 2000);
        double ta = run_a();
        (void)hipDeviceSynchronize();
        printf("A while B (extra stream %d) runs 2 x (40960 short blocks): %.0f us\n", i, ta);
    }
    int lo, hi;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t ahi, blo;
    (void)hipStreamCreateWithPriority(&ahi, hipStreamNonBlocking, hi);
    (void)hipStreamCreateWithPriority(&blo, hipStreamNonBlocking, lo);
    printf("priority range: least %d, greatest %d\n", lo, hi);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipDeviceSynchronize();
        for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(k_work, 40960, 256, 0, blo, o, 1, 2000);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_short, 120, 256, 0, ahi, p);
        (void)hipStreamSynchronize(ahi);
        double ta = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        (void)hipDeviceSynchronize();
        printf("A (high priority) while B (low priority) runs 2 x (40960 short blocks): %.0f us\n", ta);
    }
    return 0;
}
