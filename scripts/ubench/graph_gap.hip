// Does a hipGraph shorten the GPU-side gap between dependent kernels?  The same chain of 300 kernels
// (170 x 256 threads, ~6 us each, every one stamping the 100 MHz realtime counter at its first block's
// start and its last block's end) is run (a) as plain launches on one stream with the host far ahead,
// (b) as one graph captured from that stream.      hipcc --offload-arch=gfx950 -O3 graph_gap.hip -o graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned long long rt()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__global__ void k_link_like(unsigned long long *st, int launch)
{
    const unsigned long long t0 = rt();
    if (threadIdx.x == 0) atomicMin(&st[2 * launch], t0);
    while (rt() - t0 < 600) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(&st[2 * launch + 1], rt());
}
static void report(const char *name, unsigned long long *st, int N)
{
    std::vector<unsigned long long> h(2 * N);
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * N, hipMemcpyDeviceToHost);
    std::vector<double> gap;
    for (int i = 50; i + 1 < N; ++i) gap.push_back((double)(long long)(h[2 * i + 2] - h[2 * i + 1]) * 0.01);
    std::sort(gap.begin(), gap.end());
    printf("%-28s gap between dependent kernels: median %.2f us, p90 %.2f us\n", name, gap[gap.size() / 2], gap[gap.size() * 9 / 10]);
}
static void reset(unsigned long long *st, int N)
{
    std::vector<unsigned long long> h(2 * N);
    for (int i = 0; i < N; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
    (void)hipMemcpy(st, h.data(), sizeof(unsigned long long) * 2 * N, hipMemcpyHostToDevice);
}
int main()
{
    const int N = 300;
    unsigned long long *st;
    (void)hipMalloc(&st, sizeof(unsigned long long) * 2 * N);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    for (int rep = 0; rep < 2; ++rep) {
        reset(st, N);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_link_like, 170, 256, 0, s, st, i);
        (void)hipStreamSynchronize(s);
        report("stream launches", st, N);
    }
    hipGraph_t graph; hipGraphExec_t exec;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_link_like, 170, 256, 0, s, st, i);
    (void)hipStreamEndCapture(s, &graph);
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { printf("graph instantiate failed\n"); return 1; }
    for (int rep = 0; rep < 3; ++rep) {
        reset(st, N);
        (void)hipGraphLaunch(exec, s);
        (void)hipStreamSynchronize(s);
        report("hipGraph (captured chain)", st, N);
    }
    return 0;
}
