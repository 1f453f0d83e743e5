// Do matrix and vector instructions of DIFFERENT waves on one SIMD overlap on gfx950?  (round 4: k_threshold_mfma takes
// 635 SIMD-cycles per 16 x 16 tile, and 90 vector instructions x 4.5 + 7.4 MFMA x 32 = 642 -- the sum, not the maximum.)
// Each wave loops over M independent v_mfma_f32_16x16x32_f16 and V independent VALU instructions, in three orders:
// "blocked" (all M, then all V), "interleaved" (V / M vector instructions after every MFMA), and each kind alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M, int V, int ORDER, int KIND>
__global__ __launch_bounds__(256) void k_mix(float *out, int iters, float a, float b)
{
    f32x4 acc[4]; float r[8]; half8_t A, B;
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 0.5f + i;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(threadIdx.x & 3); B[i] = (_Float16)(i & 1); }
    auto valu = [&](int i) __attribute__((always_inline)) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i & 7]) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r[i & 7]) : "v"(a));
        if (KIND == 2) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(r[i & 7]) : "v"(a));
    };
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(A), "v"(B));
#pragma unroll
            for (int v = 0; v < V; ++v) valu(v);
        } else {
            constexpr int PER = M > 0 ? V / (M > 0 ? M : 1) : V;
#pragma unroll
            for (int m = 0; m < (M > 0 ? M : 1); ++m) {
                if (M > 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(A), "v"(B));
#pragma unroll
                for (int v = 0; v < PER; ++v) valu(m * PER + v);
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += r[i]; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int M, int V, int ORDER, int KIND> void run(const char *name, float *out)
{
    for (int w : {1, 2, 4}) {
        const int iters = 2000; dim3 grid(256 * w), block(256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k_mix<M, V, ORDER, KIND>), grid, block, 0, 0, out, 10, 1.0001f, 0.5f);
        hipEventRecord(e0); hipLaunchKernelGGL((k_mix<M, V, ORDER, KIND>), grid, block, 0, 0, out, iters, 1.0001f, 0.5f); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        // cycles the SIMD spends per loop iteration of ONE wave (w waves share it)
        printf("%-58s waves/SIMD=%d  %7.1f cycles per wave-iteration per SIMD\n", name, w, ms * 1e-3 * 2.4e9 / ((double)iters * w));
    }
}

int main()
{
    float *out; hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    run<8, 0, 0, 0>("8 MFMA 16x16x32 f16 alone", out);
    run<0, 64, 0, 0>("64 v_fma_f32 alone", out);
    run<8, 64, 0, 0>("8 MFMA then 64 v_fma_f32 (blocked)", out);
    run<8, 64, 1, 0>("8 x (MFMA, 8 v_fma_f32) (interleaved)", out);
    run<8, 32, 1, 0>("8 x (MFMA, 4 v_fma_f32) (interleaved)", out);
    run<0, 64, 0, 1>("64 v_cvt_pk_u8_f32 alone", out);
    run<8, 64, 0, 1>("8 MFMA then 64 v_cvt_pk_u8_f32 (blocked)", out);
    run<8, 64, 1, 1>("8 x (MFMA, 8 v_cvt_pk_u8_f32) (interleaved)", out);
    run<0, 64, 0, 2>("64 v_lshl_or_b32 alone", out);
    run<8, 64, 1, 2>("8 x (MFMA, 8 v_lshl_or_b32) (interleaved)", out);
    run<4, 64, 1, 0>("4 x (MFMA, 16 v_fma_f32) (interleaved)", out);
    run<2, 64, 1, 0>("2 x (MFMA, 32 v_fma_f32) (interleaved)", out);
    return 0;
}
