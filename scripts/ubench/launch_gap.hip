// GPU-side gap between dependent launches on one stream: each launch stamps the 100 MHz realtime
// counter when its first block begins and when its last block ends; every kernel spins ~6 us so
// that the host stays ahead.  Variants: grid size, dynamic LDS, kernarg bytes, VGPR footprint,
// dirty bytes written.    hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Big { long long v[44]; };
__device__ __forceinline__ unsigned long long rt()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ void body(unsigned long long *st, int launch, int *dirty, int dirty_n)
{
    const unsigned long long t0 = rt();
    if (threadIdx.x == 0) atomicMin(&st[2 * launch], t0);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dirty_n; i += gridDim.x * blockDim.x) dirty[i] = launch;
    while (rt() - t0 < 600) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(&st[2 * launch + 1], rt());
}
__global__ void k_plain(unsigned long long *st, int launch, int *dirty, int dirty_n) { body(st, launch, dirty, dirty_n); }
__global__ void k_lds(unsigned long long *st, int launch, int *dirty, int dirty_n)
{
    extern __shared__ int s[];
    if (launch < 0) s[threadIdx.x] = 1;
    body(st, launch, dirty, dirty_n);
}
__global__ void k_big(Big a, Big b, unsigned long long *st, int launch, int *dirty, int dirty_n)
{
    if (launch < 0) dirty[0] = (int)(a.v[launch & 31] + b.v[launch & 15]);
    body(st, launch, dirty, dirty_n);
}
__global__ void k_vgpr(unsigned long long *st, int launch, int *dirty, int dirty_n)
{
    asm volatile("v_mov_b32 v200, 0" ::: "v200");
    body(st, launch, dirty, dirty_n);
}
template <typename F> void run(const char *name, F f, unsigned long long *st)
{
    const int N = 400;
    std::vector<unsigned long long> h(2 * N);
    for (int i = 0; i < N; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
    (void)hipMemcpy(st, h.data(), sizeof(unsigned long long) * 2 * N, hipMemcpyHostToDevice);
    for (int i = 0; i < N; ++i) f(i);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * N, hipMemcpyDeviceToHost);
    std::vector<double> gap, dur;
    for (int i = 100; i + 1 < N; ++i) { gap.push_back((double)(long long)(h[2 * i + 2] - h[2 * i + 1]) * 0.01); dur.push_back((double)(h[2 * i + 1] - h[2 * i]) * 0.01); }
    std::sort(gap.begin(), gap.end()); std::sort(dur.begin(), dur.end());
    printf("%-34s gap med %.2f us p90 %.2f us   (in-kernel %.2f us)\n", name, gap[gap.size() / 2], gap[gap.size() * 9 / 10], dur[dur.size() / 2]);
}
int main()
{
    unsigned long long *st; int *dirty;
    (void)hipMalloc(&st, 1 << 16); (void)hipMalloc(&dirty, 64 << 20);
    (void)hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    Big a{}, b{};
    run("plain 1x64", [&](int i) { hipLaunchKernelGGL(k_plain, 1, 64, 0, 0, st, i, dirty, 0); }, st);
    run("plain 170x256", [&](int i) { hipLaunchKernelGGL(k_plain, 170, 256, 0, 0, st, i, dirty, 0); }, st);
    run("plain 512x256", [&](int i) { hipLaunchKernelGGL(k_plain, 512, 256, 0, 0, st, i, dirty, 0); }, st);
    run("lds 64K 170x256", [&](int i) { hipLaunchKernelGGL(k_lds, 170, 256, 64 * 1024, 0, st, i, dirty, 0); }, st);
    run("lds 150K 170x256", [&](int i) { hipLaunchKernelGGL(k_lds, 170, 256, 150 * 1024, 0, st, i, dirty, 0); }, st);
    run("kernarg 730B 170x256", [&](int i) { hipLaunchKernelGGL(k_big, 170, 256, 0, 0, a, b, st, i, dirty, 0); }, st);
    run("vgpr 201 170x256", [&](int i) { hipLaunchKernelGGL(k_vgpr, 170, 256, 0, 0, st, i, dirty, 0); }, st);
    run("dirty 256KB 170x256", [&](int i) { hipLaunchKernelGGL(k_plain, 170, 256, 0, 0, st, i, dirty, 65536); }, st);
    run("dirty 4MB 170x256", [&](int i) { hipLaunchKernelGGL(k_plain, 170, 256, 0, 0, st, i, dirty, 1 << 20); }, st);
    run("dirty 64MB 170x256", [&](int i) { hipLaunchKernelGGL(k_plain, 170, 256, 0, 0, st, i, dirty, 16 << 20); }, st);
    return 0;
}
