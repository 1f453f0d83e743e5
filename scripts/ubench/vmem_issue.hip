// What a 1-KB vector memory instruction costs the WAVE that issues it (round 5: the threshold kernel's 53 such instructions per
// 16-row step cost its waves 700 - 1 500 cycles of a 7 665-cycle step, ~300 apiece, whoever issues them -- DESIGN.md 4).
// One 1024-thread workgroup per CU.  The last NM waves issue N instructions of one form back to back (s_memtime before the
// first, behind the last = ISSUE; behind s_waitcnt vmcnt(0) = DONE); the other waves idle at a barrier or run an MFMA + VALU
// loop meanwhile, the memory waves at the same priority or above them (s_setprio 3).  Forms: global / buffer, LDS-DMA / to registers / stores, 16 / 8 / 4 bytes per lane.  Every instruction
// touches fresh bytes (HBM, as in the kernel), 1 KB contiguous per wave-instruction.
//     hipcc --offload-arch=gfx950 -O3 vmem_issue.hip -o vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int N = 16;           // instructions per timed run

enum Form { G_LDS_M0 = 0, G_LDS_FIXED, B_LDS_M0, G_LOAD, B_LOAD, G_STORE16, B_STORE16, G_STORE8, G_STORE4, G_LDS4, N_FORMS };
static const char *names[N_FORMS] = {"global_load_lds_dwordx4, m0 per piece", "global_load_lds_dwordx4, one m0", "buffer_load_dwordx4 lds, m0 per piece",
                                     "global_load_dwordx4 -> VGPR", "buffer_load_dwordx4 -> VGPR", "global_store_dwordx4", "buffer_store_dwordx4",
                                     "global_store_dwordx2 (512 B)", "global_store_dword (256 B)", "global_load_lds_dword (256 B), m0 per piece"};

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

template <int FORM>
__global__ __launch_bounds__(1024) void k(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, unsigned long long *out, int nm, int busy,
                                          int reps, int prio)
{
    extern __shared__ __align__(16) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wave - (16 - nm);
    // this CU's region: 4 MB of src and of dst, walked once
    const size_t region = (size_t)blockIdx.x * (4u << 20);
    unsigned long long issue = 0, done = 0;
    __shared__ int stop;
    if (tid == 0) stop = 0;
    __syncthreads();
    if (mi >= 0) {
        if (prio) __builtin_amdgcn_s_setprio(3);           // (the other waves stay at 0)
        uint32_t acc = 0;
        for (int rep = 0; rep < reps; ++rep) {
            const uint32_t base = (uint32_t)(((rep * nm + mi) * N) * 1024) & ((4u << 20) - 1u);
            const uint8_t *s = src + region;
            uint8_t *d = dst + region;
            const uint32_t off = base + 16u * (uint32_t)lane;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)s, 0, 4 << 20, 0x00020000);
            __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)d, 0, 4 << 20, 0x00020000);
            u32x4 v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = u32x4{(uint32_t)lane, (uint32_t)i, acc, 7u};
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long t0 = now();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t o = off + 1024u * (uint32_t)i;
                const uint32_t l = (uint32_t)(uintptr_t)lds + (uint32_t)(mi * N + i) * 1024u;
                if (FORM == G_LDS_M0) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(o), "s"(s) : "memory");
                if (FORM == G_LDS_FIXED) {
                    if (i == 0) asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(l)) : "memory");
                    asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(o), "s"(s) : "memory");
                }
                if (FORM == G_LDS4) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(base + 256u * i + 4u * lane), "s"(s) : "memory");
                if (FORM == B_LDS_M0) asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(o), "s"(rs) : "memory");
                // (loads to registers through the compiler, never through asm: it must know that the destination is written LATER -- an
                // asm load whose output it took for ready let it reuse a register of the destination as the next address, and the
                // first build of this file faulted on the data that landed there)
                if (FORM == G_LOAD) v[i] = *reinterpret_cast<const u32x4 *>(s + o);
                if (FORM == B_LOAD) v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0));
                if (FORM == G_STORE16) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(o), "v"(v[i]), "s"(d) : "memory");
                if (FORM == B_STORE16) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(v[i]), "v"(o), "s"(rd) : "memory");
                if (FORM == G_STORE8) asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(base + 512u * i + 8u * lane), "v"(u32x2{v[i][0], v[i][1]}), "s"(d) : "memory");
                if (FORM == G_STORE4) asm volatile("global_store_dword %0, %1, %2" ::"v"(base + 256u * i + 4u * lane), "v"(v[i][0]), "s"(d) : "memory");
            }
            asm volatile("" ::: "memory");
            const unsigned long long t1 = now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t2 = now();
            if (rep >= reps / 2) { issue += t1 - t0; done += t2 - t0; }
#pragma unroll
            for (int i = 0; i < N; ++i) acc += v[i][0] ^ v[i][3];
        }
        if (acc == 0x12345u) dst[0] = 1;
        if (lane == 0) {
            out[(size_t)blockIdx.x * 8 + 2 * mi] = issue / (unsigned long long)(reps - reps / 2);
            out[(size_t)blockIdx.x * 8 + 2 * mi + 1] = done / (unsigned long long)(reps - reps / 2);
        }
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) atomicAdd(&stop, 1);
    } else if (busy) {
        // the threshold kernel's kind of work: MFMA 16x16x32 f16 chains with vector instructions between them
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane + i); b[i] = (_Float16)(0.001f * i); }
        float x = (float)lane;
        while (__hip_atomic_load(&stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nm) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
                x = __builtin_fmaf(x, 1.0001f, 0.5f); x = __builtin_fmaf(x, 0.9999f, -0.5f);
                x = __builtin_fmaf(x, 1.0001f, 0.5f); x = __builtin_fmaf(x, 0.9999f, -0.5f);
            }
        }
        if (c[0] + x == 12345.678f) dst[1] = 2;
    }
}

template <int FORM> void run(const uint8_t *src, uint8_t *dst, unsigned long long *out, int nm, int busy, int prio)
{
    const int grid = 256, reps = 12;
    (void)hipFuncSetAttribute((const void *)k<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipMemset(out, 0, sizeof(unsigned long long) * grid * 8);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<FORM>, dim3(grid), dim3(1024), 64 * 1024, 0, src, dst, out, nm, busy, reps, prio);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 8);
    (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * grid * 8, hipMemcpyDeviceToHost);
    std::vector<double> is, dn;
    for (int b = 0; b < grid; ++b)
        for (int m = 0; m < nm; ++m) { is.push_back((double)h[b * 8 + 2 * m] / N); dn.push_back((double)h[b * 8 + 2 * m + 1] / N); }
    std::sort(is.begin(), is.end()); std::sort(dn.begin(), dn.end());
    printf("%-44s %d memory wave%s%s, others %-5s  issue %6.0f cycles per instruction (p90 %6.0f)   issue + drain %6.0f\n", names[FORM], nm, nm > 1 ? "s" : " ", prio ? " at priority 3" : "",
           busy ? "busy" : "idle", is[is.size() / 2], is[is.size() * 9 / 10], dn[dn.size() / 2]);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    uint8_t *src, *dst; unsigned long long *out;
    (void)hipMalloc(&src, (size_t)256 * (4u << 20)); (void)hipMalloc(&dst, (size_t)256 * (4u << 20)); (void)hipMalloc(&out, 1 << 16);
    (void)hipMemset(src, 1, (size_t)256 * (4u << 20)); (void)hipMemset(dst, 0, (size_t)256 * (4u << 20));
    for (int busy = 0; busy < 2; ++busy)
        for (int prio = 0; prio <= busy; ++prio)
            for (int nm : {1, 4}) {
                run<G_LDS_M0>(src, dst, out, nm, busy, prio);
                run<G_LDS_FIXED>(src, dst, out, nm, busy, prio);
                run<B_LDS_M0>(src, dst, out, nm, busy, prio);
                run<G_LDS4>(src, dst, out, nm, busy, prio);
                run<G_LOAD>(src, dst, out, nm, busy, prio);
                run<B_LOAD>(src, dst, out, nm, busy, prio);
                run<G_STORE16>(src, dst, out, nm, busy, prio);
                run<B_STORE16>(src, dst, out, nm, busy, prio);
                run<G_STORE8>(src, dst, out, nm, busy, prio);
                run<G_STORE4>(src, dst, out, nm, busy, prio);
            }
    return 0;
}
