// When does ONE fat workgroup (1024 threads, L KB of LDS: k_link) get a compute unit while another stream's resident grid
// holds the chip?  Filler: B blocks x 256 threads that spin ~150 us each, with F KB of LDS per block (k_windows: 10 KB,
// 2048 blocks; k_geometry: 26 KB; k_threshold_strip: 12.5 KB, 768 blocks of 128-VGPR waves).  20 us after the filler is
// launched (the host sees its first stamp, then launches) the probe goes out on another stream; reported: the probe's
// first instruction relative to the filler's (device clock; ~20-30 us when the probe is placed at once, ~150+ when it has
// to wait for filler blocks to finish), median of 7 repetitions.  (Row "none": relative to a one-wave kernel launched
// just before the probe.)
//     hipcc --offload-arch=gfx950 -O3 placement.hip -o placement
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned long long rt()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
extern __shared__ unsigned int s_dyn[];
template <int VGPRS>
__global__ __launch_bounds__(256) void k_filler(unsigned long long *t_start, int ticks, float *sink)
{
    float keep[VGPRS > 64 ? 96 : 8];
#pragma unroll
    for (int i = 0; i < (VGPRS > 64 ? 96 : 8); ++i) keep[i] = threadIdx.x * 0.5f + i;
    if (threadIdx.x == 0) s_dyn[0] = blockIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) *t_start = rt();
    const unsigned long long t0 = rt();
    while (rt() - t0 < (unsigned long long)ticks) {
#pragma unroll
        for (int i = 0; i < (VGPRS > 64 ? 96 : 8); ++i) keep[i] = __builtin_fmaf(keep[i], 1.0000001f, 0.25f);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < (VGPRS > 64 ? 96 : 8); ++i) acc += keep[i];
    if (acc == 12345.f) sink[0] = acc + s_dyn[0];
}
__global__ __launch_bounds__(1024) void k_probe(unsigned long long *t_first)
{
    if (threadIdx.x == 0) { s_dyn[0] = 1; *t_first = rt(); }
}
// the same with ~80 VGPRs per lane (k_link's allocation)
__global__ __launch_bounds__(1024) void k_probe_fat(unsigned long long *t_first, float *sink)
{
    if (threadIdx.x == 0) { s_dyn[0] = 1; *t_first = rt(); }
    float keep[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) keep[i] = threadIdx.x * 0.25f + i;
    for (int it = 0; it < 4; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) keep[i] = __builtin_fmaf(keep[i], 1.0000001f, keep[(i + 7) & 63]);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) acc += keep[i];
    if (acc == 12345.f) sink[0] = acc;
}
// k_track's shape: many 256-thread workgroups of ~86-VGPR waves that live ~10 us
__global__ __launch_bounds__(256) void k_many(float *sink, int ticks)
{
    float keep[72];
#pragma unroll
    for (int i = 0; i < 72; ++i) keep[i] = threadIdx.x * 0.25f + i;
    const unsigned long long t0 = rt();
    while (rt() - t0 < (unsigned long long)ticks) {
#pragma unroll
        for (int i = 0; i < 72; ++i) keep[i] = __builtin_fmaf(keep[i], 1.0000001f, keep[(i + 5) % 72]);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 72; ++i) acc += keep[i];
    if (acc == 12345.f) sink[0] = acc;
}
__global__ void k_stamp(unsigned long long *t) { if (threadIdx.x == 0) *t = rt(); }
__global__ void k_delay(int ticks) { const unsigned long long t0 = rt(); while (rt() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8); }

int main()
{
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 64);
    hipStream_t a, b;
    (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    (void)hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_filler<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)k_filler<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    struct Case { const char *name; int blocks, lds_kb, vgprs; };
    const Case fillers[] = {{"none", 0, 0, 48}, {"2048 x 10 KB, 48 VGPRs (k_windows)", 2048, 10, 48}, {"1024 x 10 KB, 48 VGPRs", 1024, 10, 48},
                            {"768 x 10 KB, 48 VGPRs", 768, 10, 48}, {"1536 x 26 KB, 128 VGPRs (k_geometry)", 1536, 26, 128},
                            {"768 x 26 KB, 128 VGPRs", 768, 26, 128}, {"768 x 12 KB, 128 VGPRs (k_threshold_strip)", 768, 12, 128},
                            {"512 x 12 KB, 128 VGPRs", 512, 12, 128}, {"1024 x 0 KB, 48 VGPRs", 1024, 0, 48}, {"2048 x 0 KB, 48 VGPRs", 2048, 0, 48}};
    const int probe_lds[] = {1, 32, 64, 96, 131};
    printf("probe: 1 workgroup x 1024 threads; its first instruction, us after the filler's first instruction (the filler's blocks run 150 us)\n");
    printf("%-46s", "filler \\ probe LDS");
    for (int l : probe_lds) printf("%8d KB", l);
    printf("\n");
    for (const Case &f : fillers) {
        printf("%-46s", f.name);
        double small_us = 0.0;
        for (int l : probe_lds) {
            std::vector<double> us;
            for (int rep = 0; rep < 7; ++rep) {
                (void)hipDeviceSynchronize();
                (void)hipMemset(d, 0, 24);
                (void)hipDeviceSynchronize();
                unsigned long long *dd = d;
                // the probe's stream first sleeps 40 us on one wave (so that the filler, launched right after it on the
                // other stream, holds the chip when the probe is dispatched), stamps, then launches the probe
                hipLaunchKernelGGL(k_delay, 1, 64, 0, b, 4000);
                if (f.blocks) {
                    if (f.vgprs > 64) hipLaunchKernelGGL(k_filler<128>, f.blocks, 256, f.lds_kb * 1024, a, dd, 15000, sink);
                    else hipLaunchKernelGGL(k_filler<48>, f.blocks, 256, f.lds_kb * 1024, a, dd, 15000, sink);
                }
                hipLaunchKernelGGL(k_stamp, 1, 64, 0, b, dd + 1);
                hipLaunchKernelGGL(k_probe, 1, 1024, l * 1024, b, dd + 2);
                (void)hipDeviceSynchronize();
                unsigned long long t[3];
                (void)hipMemcpy(t, d, 24, hipMemcpyDeviceToHost);
                const unsigned long long t0 = t[0], t1 = t[1], t2 = t[2];
                us.push_back(f.blocks ? (double)(t2 - t0) / 100.0 : (double)(t2 - t1) / 100.0);
                if (rep == 6 && l == 1 && f.blocks) small_us = (double)(t1 - t0) / 100.0;
                (void)hipDeviceSynchronize();
            }
            std::sort(us.begin(), us.end());
            printf("%11.1f", us[us.size() / 2]);
        }
        printf("   (one 64-thread workgroup launched just before the probe: %.1f)\n", small_us);
    }
    // ---- the pipeline's situation: the probe's stream runs a k_track-like kernel (1288 x 256 threads, ~10 us) that is
    // still running when the filler starts, then the fat probe (80 VGPRs, 101 KB)
    (void)hipFuncSetAttribute((const void *)k_probe_fat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("\nfat probe (1024 threads, ~80 VGPRs, 101 KB LDS) behind a 1288 x 256-thread kernel that overlaps the filler's start; us after the filler's first instruction\n");
    for (const Case &f : fillers) {
        if (!f.blocks) continue;
        std::vector<double> us;
        for (int rep = 0; rep < 7; ++rep) {
            (void)hipMemset(d, 0, 24);
            (void)hipDeviceSynchronize();
            hipLaunchKernelGGL(k_delay, 1, 64, 0, b, 1000);
            hipLaunchKernelGGL(k_many, 1288, 256, 0, b, sink, 1500);
            hipLaunchKernelGGL(k_delay, 1, 64, 0, a, 1800);          // the filler starts ~8 us into k_many
            if (f.vgprs > 64) hipLaunchKernelGGL(k_filler<128>, f.blocks, 256, f.lds_kb * 1024, a, d, 15000, sink);
            else hipLaunchKernelGGL(k_filler<48>, f.blocks, 256, f.lds_kb * 1024, a, d, 15000, sink);
            hipLaunchKernelGGL(k_stamp, 1, 64, 0, b, d + 1);
            hipLaunchKernelGGL(k_probe_fat, 1, 1024, 101 * 1024, b, d + 2, sink);
            (void)hipDeviceSynchronize();
            unsigned long long t[3];
            (void)hipMemcpy(t, d, 24, hipMemcpyDeviceToHost);
            us.push_back(((double)t[2] - (double)t[0]) / 100.0);
        }
        std::sort(us.begin(), us.end());
        printf("%-46s min %8.1f  median %8.1f  max %8.1f\n", f.name, us.front(), us[us.size() / 2], us.back());
    }
    return 0;
}
