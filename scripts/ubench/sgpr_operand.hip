// Does a scalar-register (or literal) source operand change the issue cost of a full-rate f32 VALU instruction
// on gfx950?  Cycles per wave-instruction per SIMD for v_fmac_f32 / v_fma_f32 / v_add_f32 with all-VGPR sources
// and with one SGPR / literal source, 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(float *out, int iters, float a, float b)
{
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 0.5f + i;
    float va = a + threadIdx.x * 1e-9f, vb = b + threadIdx.x * 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(va), "v"(vb));
                if (KIND == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "s"(a), "v"(vb));
                if (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(va), "v"(vb));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "s"(a), "v"(vb));
                if (KIND == 4) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[i]) : "v"(va));
                if (KIND == 5) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[i]) : "s"(a));
                if (KIND == 6) asm volatile("v_add_f32 %0, 0x4b400000, %0" : "+v"(r[i]));
                if (KIND == 7) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(va), "v"(r[(i + 2) & 7]));   // three different VGPRs
                if (KIND == 8) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char *name, float *out)
{
    for (int w : {1, 2, 4}) {
        int iters = 1000; dim3 grid(256 * w), block(256);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, 10, 1.0001f, 0.5f);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s waves/SIMD=%d  %.2f cyc/wave-instr/SIMD (at 2.4 GHz)\n", name, w, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * w));
    }
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 4 * 1024 * 64 * sizeof(float));
    run<0>("v_fmac_f32 v, v, v", out); run<1>("v_fmac_f32 v, s, v", out); run<2>("v_fma_f32 v, v, v, v", out); run<3>("v_fma_f32 v, s, v, v", out);
    run<4>("v_add_f32 v, v, v", out); run<5>("v_add_f32 v, s, v", out); run<6>("v_add_f32 v, literal, v", out);
    run<7>("v_fma_f32 three distinct VGPRs", out); run<8>("v_fmac_f32 distinct VGPRs", out);
    return 0;
}
