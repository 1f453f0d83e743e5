// Packed f32 VALU on gfx950: issue cost of v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (two f32 results per lane)
// next to v_fma_f32, with a broadcast (op_sel_hi) weight, an SGPR-pair weight, and in dependent chains of the
// shape the 11-tap filter has.  Cycles per wave-instruction per SIMD, 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float *out, int iters, float a, float b)
{
    float2v r[8];
    for (int i = 0; i < 8; ++i) { r[i].x = threadIdx.x * 0.5f + i; r[i].y = threadIdx.x * 0.25f - i; }
    float2v va = {a + threadIdx.x * 1e-9f, a - threadIdx.x * 1e-9f}, vb = {b + threadIdx.x * 1e-9f, b};
    float2v sa = {a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(va), "v"(vb));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(r[i]) : "v"(va), "v"(vb));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(va), "s"(sa));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(r[i]) : "v"(va));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(va));
                if (KIND == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i].x) : "v"(va.x), "v"(vb.x));
                // one dependent chain per wave: every instruction reads the previous result
                if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(va), "v"(vb));
                if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0].x) : "v"(va.x), "v"(vb.x));
                // two interleaved chains
                if (KIND == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[i & 1]) : "v"(va), "v"(vb));
                if (KIND == 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i & 1].x) : "v"(va.x), "v"(vb.x));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;
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
        printf("%-44s waves/SIMD=%d  %.2f cyc/wave-instr/SIMD (at 2.4 GHz)\n", name, w, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * w));
    }
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 4 * 1024 * 64 * sizeof(float));
    run<0>("v_pk_fma_f32 v, v, v", out); run<1>("v_pk_fma_f32 v, v(lo broadcast), v", out); run<2>("v_pk_fma_f32 v, s[2], v", out);
    run<3>("v_pk_add_f32", out); run<4>("v_pk_mul_f32", out); run<5>("v_fma_f32", out);
    run<6>("v_pk_fma_f32 one dependent chain", out); run<7>("v_fma_f32 one dependent chain", out);
    run<8>("v_pk_fma_f32 two chains", out); run<9>("v_fma_f32 two chains", out);
    return 0;
}
