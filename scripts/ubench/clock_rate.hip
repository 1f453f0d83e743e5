// What do s_memtime / s_memrealtime / a dependent-FMA chain tick at?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *out, float *sink, int iters)
{
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);   // dependent chain
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    sink[threadIdx.x] = x;
}
int main()
{
    unsigned long long *o, h[2]; float *s;
    (void)hipMalloc(&o, 16); (void)hipMalloc(&s, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int iters : {100000, 1000000, 4000000}) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, 1, 64, 0, 0, o, s, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
            printf("iters %8d: %.3f ms  memtime %llu (%.1f MHz)  realtime %llu (%.1f MHz)  memtime ticks/iter %.3f\n", iters, ms,
                   h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), (double)h[0] / iters);
        }
    }
    return 0;
}
