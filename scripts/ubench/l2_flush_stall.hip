// Does the end of a kernel that wrote a lot of data (stream B) stall short dependent kernels on
// stream A?  A: 400 x (128 blocks; each block loads a word the previous launch wrote, then stores
// one).  B: one resident kernel writing `mb` MB, either with plain stores, nontemporal stores, or
// write-through (sc0 sc1) stores.  Prints A's per-launch period histogram around B's end.
//     hipcc --offload-arch=gfx950 -O3 l2_flush_stall.hip -o l2_flush_stall
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
__global__ void k_chain(int *state, unsigned long long *stamps, int launch)
{
    const unsigned long long t0 = rt();
    int v = state[(launch & 1) * 4096 + blockIdx.x * 16];            // written by the previous launch (another XCD's L2)
    v += state[(launch & 1) * 4096 + ((blockIdx.x + 37) & 127) * 16];
    state[((launch + 1) & 1) * 4096 + blockIdx.x * 16] = v + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[2 * launch] = t0; stamps[2 * launch + 1] = rt(); }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_writer(uint4 *dst, size_t n16, unsigned long long *when)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) when[0] = rt();
    const uint4 val = make_uint4(1, 2, 3, threadIdx.x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (MODE == 0) dst[i] = val;
        else if (MODE == 1) {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store((v4u){val.x, val.y, val.z, val.w}, reinterpret_cast<v4u *>(dst + i));
        }
        else {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u q = {val.x, val.y, val.z, val.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(q) : "memory");
        }
    }
    if (threadIdx.x == 0) atomicMax(&when[1], rt());
}
int main()
{
    int *state; unsigned long long *stamps, *when; uint4 *dst;
    const size_t mb = 72, n16 = mb * 1024 * 1024 / 16;
    (void)hipMalloc(&state, 2 * 4096 * 4); (void)hipMemset(state, 0, 2 * 4096 * 4);
    (void)hipMalloc(&stamps, 8 * 2 * 1024); (void)hipMalloc(&when, 16); (void)hipMalloc(&dst, n16 * 16);
    hipStream_t a, b;
    (void)hipStreamCreate(&a); (void)hipStreamCreate(&b);
    const char *names[3] = {"plain stores", "nontemporal stores", "write-through (sc0 sc1) stores"};
    for (int trial = 0; trial < 9; ++trial) {
        const int mode = trial % 3;
        (void)hipMemset(when, 0, 16);
        (void)hipDeviceSynchronize();
        const int N = 400;
        for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_chain, 128, 64, 0, a, state, stamps, i);   // get going
        if (mode == 0) hipLaunchKernelGGL(k_writer<0>, 768, 256, 0, b, dst, n16, when);
        if (mode == 1) hipLaunchKernelGGL(k_writer<1>, 768, 256, 0, b, dst, n16, when);
        if (mode == 2) hipLaunchKernelGGL(k_writer<2>, 768, 256, 0, b, dst, n16, when);
        for (int i = 40; i < N; ++i) hipLaunchKernelGGL(k_chain, 128, 64, 0, a, state, stamps, i);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * N), w(2);
        (void)hipMemcpy(h.data(), stamps, 16 * N, hipMemcpyDeviceToHost);
        (void)hipMemcpy(w.data(), when, 16, hipMemcpyDeviceToHost);
        double worst = 0, worst_at = 0; int slow = 0;
        std::vector<double> per;
        for (int i = 41; i + 1 < N; ++i) {
            double p = (double)(h[2 * i + 2] - h[2 * i]) * 0.01;
            per.push_back(p);
            if (p > worst) { worst = p; worst_at = ((double)h[2 * i] - (double)w[1]) * 0.01; }
            if (p > 8.0) ++slow;
        }
        std::sort(per.begin(), per.end());
        // periods of the chain launches that began within [-30, +30] us of the writer's end
        double near_worst = 0; int near_n = 0;
        for (int i = 41; i + 1 < N; ++i) {
            const double at = ((double)h[2 * i] - (double)w[1]) * 0.01, p = (double)(h[2 * i + 2] - h[2 * i]) * 0.01;
            if (at > -30 && at < 30) { ++near_n; near_worst = std::max(near_worst, p); }
        }
        printf("%-32s writer ran %.1f us | chain period median %.2f us | worst within +-30 us of the writer's end: %.1f us (%d launches) | worst overall %.1f us at %+.0f us\n",
               names[mode], (double)(w[1] - w[0]) * 0.01, per[per.size() / 2], near_worst, near_n, worst, worst_at);
    }
    return 0;
}
