// VALU issue-rate microbenchmark for gfx950: time R iterations of 64 independent instructions of
// one kind with W waves per SIMD resident; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef short short2_t __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(float *out, int iters, float a, float b, unsigned ua, unsigned ub)
{
    float r[8]; float2_t p[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x * 0.5f + i; p[i] = float2_t{r[i], r[i] + 1}; u[i] = threadIdx.x * 17u + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(float2_t{a, a}), "v"(float2_t{b, b}));
                if (KIND == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 3) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r[i]) : "v"(u[i]));
                if (KIND == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 5) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
                if (KIND == 6) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                if (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(float2_t{a, a}));
                if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 10) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(ua));
                if (KIND == 11) asm volatile("v_rndne_f32 %0, %0" : "+v"(r[i]));
                if (KIND == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += r[i] + p[i].x + p[i].y + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char *name, int waves_per_simd)
{
    float *out; hipMalloc(&out, 256 * 4 * 1024 * 64 * sizeof(float));
    int iters = 2000; dim3 grid(256 * waves_per_simd), block(256);  // 4 waves per block -> 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, 10, 1.0001f, 0.5f, 0x03020100u, 0x07060504u);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f, 0x03020100u, 0x07060504u); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 64 * waves_per_simd;   // wave-instructions issued on each SIMD
    printf("%-16s waves/SIMD=%d  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w); run<12>("v_fmac_f32", w); run<1>("v_pk_fma_f32", w); run<7>("v_add_f32", w); run<8>("v_pk_add_f32", w);
        run<2>("v_perm_b32", w); run<3>("v_cvt_f32_ubyte1", w); run<4>("v_add3_u32", w); run<5>("v_pk_add_u16", w);
        run<6>("v_mov_dpp wshr", w); run<9>("v_mov_b32", w); run<10>("v_lshl_add_u32", w); run<11>("v_rndne_f32", w);
    }
    return 0;
}
