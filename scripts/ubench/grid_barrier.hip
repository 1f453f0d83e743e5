// Cost of a software grid barrier among resident blocks (what a persistent link kernel would pay per
// frame instead of a kernel boundary).  Flat: one counter.  Tree: 8 sub-counters + a top counter.
//     hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long rt()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void k_flat(unsigned *ctr, int iters, unsigned long long *out, int *err)
{
    const unsigned long long t0 = rt();
    for (int it = 1; it <= iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(ctr, 1u);
            unsigned spins = 0;
            while (ld(ctr) < (unsigned)it * gridDim.x) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { atomicOr(err, 1); break; }
            }
            __threadfence();
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = rt() - t0;
}
// sub-counters are 128 bytes apart; the block that completes a sub-counter bumps the top counter
__global__ void k_tree(unsigned *ctr, int iters, unsigned long long *out, int *err)
{
    const unsigned long long t0 = rt();
    const unsigned g = blockIdx.x & 7u;
    const unsigned members = (gridDim.x - g + 7u) / 8u;
    unsigned *sub = ctr + 32 * (1 + g), *top = ctr;
    for (int it = 1; it <= iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(sub, 1u) == (unsigned)it * members - 1u) atomicAdd(top, 1u);
            unsigned spins = 0;
            while (ld(top) < (unsigned)it * 8u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { atomicOr(err, 1); break; }
            }
            __threadfence();
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = rt() - t0;
}
// Round 5 (the 4K link's paper): a barrier among FEW workgroups -- 7 or 8, what 5 000 tracks at 768 seats
// need -- placed on ONE XCD (launch 8 x as many and let blockIdx % 8 != 0 leave: round-robin dealing puts the rest on one XCD,
// for speed only), with the fences a link frame needs (release before the arrive, acquire after the wait) or with none (the
// frame's exchange done through agent-scope atomics themselves), 768 threads per workgroup like k_batch.
template <bool FENCES, bool ONE_XCD>
__global__ void k_few(unsigned *ctr, int iters, unsigned long long *out, int *err, unsigned members)
{
    if (ONE_XCD && (blockIdx.x & 7u) != 0u) return;
    const unsigned long long t0 = rt();
    for (int it = 1; it <= iters; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (threadIdx.x == 0) {
            if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (ld(ctr) < (unsigned)it * members) {
                if (++spins > 4000000u) { atomicOr(err, 1); break; }
            }
            if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = rt() - t0;
}

int main()
{
    unsigned *ctr; unsigned long long *out, h; int *err, herr;
    (void)hipMalloc(&ctr, 4096); (void)hipMalloc(&out, 8); (void)hipMalloc(&err, 4);
    const int iters = 2000;
    for (int blocks : {16, 64, 128, 256, 512}) {
        for (int tree = 0; tree < 2; ++tree) {
            (void)hipMemset(ctr, 0, 4096); (void)hipMemset(err, 0, 4);
            if (tree) hipLaunchKernelGGL(k_tree, blocks, 256, 0, 0, ctr, iters, out, err);
            else hipLaunchKernelGGL(k_flat, blocks, 256, 0, 0, ctr, iters, out, err);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            printf("%3d blocks %s: %.2f us per barrier%s\n", blocks, tree ? "tree" : "flat", h * 0.01 / iters, herr ? "  (TIMEOUT)" : "");
        }
    }
    for (unsigned members : {2u, 4u, 7u, 8u}) {
        for (int mode = 0; mode < 4; ++mode) {
            const bool fences = mode & 1, one = mode & 2;
            (void)hipMemset(ctr, 0, 4096); (void)hipMemset(err, 0, 4);
            const unsigned grid = one ? 8 * members : members;
            if (fences && one) hipLaunchKernelGGL((k_few<true, true>), grid, 768, 0, 0, ctr, iters, out, err, members);
            else if (fences) hipLaunchKernelGGL((k_few<true, false>), grid, 768, 0, 0, ctr, iters, out, err, members);
            else if (one) hipLaunchKernelGGL((k_few<false, true>), grid, 768, 0, 0, ctr, iters, out, err, members);
            else hipLaunchKernelGGL((k_few<false, false>), grid, 768, 0, 0, ctr, iters, out, err, members);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            printf("%u workgroups of 768 threads, %s, %s: %.2f us per barrier%s\n", members, one ? "one XCD" : "dealt over the XCDs",
                   fences ? "release + acquire fences" : "counter only", h * 0.01 / iters, herr ? "  (TIMEOUT)" : "");
        }
    }
    return 0;
}
