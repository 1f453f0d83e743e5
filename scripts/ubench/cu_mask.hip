// Compute-unit masks on a stream (hipExtStreamCreateWithCUMask): which units does a kernel land on, and does a fat workgroup
// on ANOTHER stream start at once when the masked stream's resident grid holds every unit it may use?  (Round 5: at 4K the
// two-launch link's k_link -- 1024 threads -- takes 10.5 us alone and 22 - 38 us beside k_windows / k_geometry, whose resident
// grids hold every wave slot; profiles/r05_timeline_4k.txt.)
//   part 1: 2048 blocks of 256 threads that spin 30 us record (XCC_ID, HW_ID) -> distinct units used, per XCC, under masks
//   part 2: filler (2048 blocks x 256 threads, 10 KB of LDS, 150 us) on a masked stream, then a 1024-thread probe on an
//           unmasked stream 20 us later: the probe's first instruction relative to the filler's
//     hipcc --offload-arch=gfx950 -O3 cu_mask.hip -o cu_mask
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
__device__ __forceinline__ unsigned long long rt()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
extern __shared__ unsigned int s_dyn[];
__global__ __launch_bounds__(256) void k_where(uint32_t *out, int ticks, unsigned long long *t_start)
{
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        s_dyn[0] = hw;
        if (blockIdx.x == 0 && t_start) *t_start = rt();
    }
    const unsigned long long t0 = rt();
    while (rt() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(4);
}
__global__ __launch_bounds__(1024) void k_probe(unsigned long long *t_first, uint32_t *where)
{
    if (threadIdx.x == 0) {
        s_dyn[0] = 1; *t_first = rt();
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[0] = hw; where[1] = xcc;
    }
}
__global__ __launch_bounds__(256) void k_probe_small(unsigned long long *t_first, uint32_t *where)
{
    if (threadIdx.x == 0) {
        s_dyn[0] = 1; *t_first = rt();
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[0] = hw; where[1] = xcc;
    }
}
// gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
static uint32_t unit_of(uint32_t hw, uint32_t xcc) { return ((xcc & 15u) << 12) | (((hw >> 13) & 7u) << 8) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u); }

static void where_run(hipStream_t st, const char *name, uint32_t *d_out)
{
    const int blocks = 2048;
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 10 * 1024, st, d_out, 3000, (unsigned long long *)nullptr);
    (void)hipStreamSynchronize(st);
    std::vector<uint32_t> h(2 * blocks);
    (void)hipMemcpy(h.data(), d_out, sizeof(uint32_t) * 2 * blocks, hipMemcpyDeviceToHost);
    std::set<uint32_t> units; std::map<uint32_t, std::set<uint32_t>> per_xcc;
    for (int b = 0; b < blocks; ++b) { const uint32_t u = unit_of(h[2 * b], h[2 * b + 1]); units.insert(u); per_xcc[h[2 * b + 1] & 15u].insert(u & 0xFFFu); }
    printf("%-44s %3zu units:", name, units.size());
    for (auto &kv : per_xcc) printf("  xcc%u %zu", kv.first, kv.second.size());
    printf("\n");
    static std::set<uint32_t> all;
    if (all.empty()) all = units;        // (the first run: every unit)
    else if (units.size() < all.size() && all.size() - units.size() <= 40) {
        printf("    missing (xcc se sh cu):");
        for (uint32_t u : all) if (!units.count(u)) printf(" %x.%x.%x.%x", u >> 12, (u >> 8) & 15u, (u >> 4) & 15u, u & 15u);
        printf("\n");
    }
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    printf("device: %s, %d compute units\n", prop.name, prop.multiProcessorCount);
    uint32_t *d_out; unsigned long long *d_t; uint32_t *d_w;
    (void)hipMalloc(&d_out, sizeof(uint32_t) * 2 * 4096); (void)hipMalloc(&d_t, 64); (void)hipMalloc(&d_w, 64);
    (void)hipFuncSetAttribute((const void *)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    where_run(nullptr, "default stream", d_out);
    struct M { const char *name; std::vector<uint32_t> m; };
    std::vector<M> masks;
    { std::vector<uint32_t> m(8, 0xFFFFFFFFu); masks.push_back({"mask: 256 bits set", m}); }
    { std::vector<uint32_t> m(8, 0xFFFFFFFFu); m[0] = 0xFFFFFF00u; masks.push_back({"mask: bits 0..7 clear", m}); }
    { std::vector<uint32_t> m(8, 0xFFFFFFFFu); m[7] = 0x00FFFFFFu; masks.push_back({"mask: bits 248..255 clear", m}); }
    { std::vector<uint32_t> m(8, 0xFFFFFFFFu); for (int i = 0; i < 8; ++i) m[i] = 0xFFFFFFFEu; masks.push_back({"mask: every 32nd bit clear (8 bits)", m}); }
    { std::vector<uint32_t> m(8, 0u); m[0] = 0xFFFFu; masks.push_back({"mask: bits 0..15 only", m}); }
    { std::vector<uint32_t> m(8, 0xFFFFFFFFu); m[0] = 0u; masks.push_back({"mask: bits 0..31 clear", m}); }
    { std::vector<uint32_t> m(8, 0xFFFFFFFFu); m[0] = 0xFFFF0000u; masks.push_back({"mask: bits 0..15 clear", m}); }
    { std::vector<uint32_t> m(8, 0xFFFFFFFFu); m[0] = 0xFFFFFF00u; m[2] = 0xFFFFFF00u; m[4] = 0xFFFFFF00u; m[6] = 0xFFFFFF00u; masks.push_back({"mask: bits 0..7, 64..71, 128..135, 192..199 clear", m}); }
    { std::vector<uint32_t> m(1, 0xFFFFFF00u); masks.push_back({"mask: 32 bits, 0..7 clear", m}); }
    { std::vector<uint32_t> m(2, 0xFFFFFFFFu); m[0] = 0xFFFFFFF0u; masks.push_back({"mask: 64 bits, 0..3 clear", m}); }
    std::vector<hipStream_t> streams;
    for (auto &mk : masks) {
        hipStream_t s = nullptr;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mk.m.size(), mk.m.data());
        if (e != hipSuccess) { printf("%-44s hipExtStreamCreateWithCUMask: %s\n", mk.name, hipGetErrorString(e)); streams.push_back(nullptr); continue; }
        streams.push_back(s);
        where_run(s, mk.name, d_out);
    }
    // part 2: the probe's start beside a resident filler, filler on: the default stream / each masked stream
    hipStream_t probe_st; (void)hipStreamCreateWithFlags(&probe_st, hipStreamNonBlocking);
    unsigned long long *h_t; (void)hipHostMalloc(&h_t, 64);
    int probe_kind = 0;       // 0: 1024 threads + 64 KB of LDS; 1: 256 threads + 1 KB; 2: 20 workgroups of 256 threads
    auto placement = [&](hipStream_t fill_st, const char *name) {
        std::vector<double> d;
        for (int rep = 0; rep < 7; ++rep) {
            (void)hipMemset(d_t, 0, 64);
            hipLaunchKernelGGL(k_where, dim3(2048), dim3(256), 10 * 1024, fill_st, d_out, 15000, d_t);
            h_t[0] = 0;
            for (int poll = 0; poll < 100000 && h_t[0] == 0; ++poll) {      // (bounded: a filler that never starts ends the run)
                (void)hipMemcpyAsync(h_t, d_t, 8, hipMemcpyDeviceToHost, probe_st);
                (void)hipStreamSynchronize(probe_st);
            }
            if (h_t[0] == 0) { printf("the filler on %s did not start\n", name); (void)hipDeviceSynchronize(); return; }
            if (probe_kind == 0) hipLaunchKernelGGL(k_probe, dim3(1), dim3(1024), 64 * 1024, probe_st, d_t + 1, d_w);
            else hipLaunchKernelGGL(k_probe_small, dim3(probe_kind == 2 ? 20 : 1), dim3(256), 1024, probe_st, d_t + 1, d_w);
            (void)hipDeviceSynchronize();
            unsigned long long t[2]; (void)hipMemcpy(t, d_t, 16, hipMemcpyDeviceToHost);
            d.push_back((double)(long long)(t[1] - t[0]) * 0.01);
        }
        std::sort(d.begin(), d.end());
        uint32_t w[2]; (void)hipMemcpy(w, d_w, 8, hipMemcpyDeviceToHost);
        printf("probe beside a filler on %-36s first instruction %7.1f us after the filler's (median of 7; min %.1f max %.1f); last probe on unit %03x\n", name, d[3], d[0], d[6],
               unit_of(w[0], w[1]));
    };
    (void)hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    placement(nullptr, "the default stream:");
    for (size_t i = 0; i < masks.size(); ++i)
        if (streams[i]) placement(streams[i], masks[i].name);
    // how many units per XCC must stay free, and where: k units of SE0 (bits 8 x + 32 c, c < k), or one unit in each of the first k SEs
    for (probe_kind = 0; probe_kind < 3; ++probe_kind) {
        printf("probe: %s\n", probe_kind == 0 ? "1 workgroup of 1024 threads, 64 KB of LDS" : probe_kind == 1 ? "1 workgroup of 256 threads" : "20 workgroups of 256 threads");
        for (int shape = 0; shape < 2; ++shape)
            for (int k = 1; k <= 4; ++k) {
                std::vector<uint32_t> m(8, 0xFFFFFFFFu);
                for (int c = 0; c < k; ++c)
                    for (int x = 0; x < 8; ++x) {
                        const int bit = shape == 0 ? x + 8 * (4 * c) : x + 8 * c;      // j = 4 c: SE0's unit c; j = c: SE c's first unit
                        m[bit >> 5] &= ~(1u << (bit & 31));
                    }
                hipStream_t st = nullptr;
                if (hipExtStreamCreateWithCUMask(&st, 8, m.data()) != hipSuccess) continue;
                char name[96];
                snprintf(name, sizeof name, shape == 0 ? "free: %d unit(s) of SE0 per XCC" : "free: the first unit of %d SE(s) per XCC", k);
                placement(st, name);
                (void)hipStreamDestroy(st);
            }
    }
    return 0;
}
