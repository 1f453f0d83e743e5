// Probe for the threshold kernel's matrix-pipe variant (gfx950):
//   A. issue cost of the VALU instructions such a kernel is made of (cycles per wave-instruction per SIMD),
//   B. cycles per MFMA of the candidate shapes, alone and with N independent v_fma_f32 per MFMA in the
//      same wave (how much vector work hides behind a matrix instruction),
//   C. the operand / accumulator lane maps of v_mfma_f32_32x32x16_f16, checked with exact integers,
//      including an accumulator tile reused as the B operand of the next MFMA (k permutation).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- A: VALU issue cost
template <int KIND>
__global__ void k_valu(float *out, int iters, float a, float b, unsigned ua, unsigned ub)
{
    float r[8]; unsigned u[8]; unsigned long long q[4];
    for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x * 0.5f + i; u[i] = threadIdx.x * 17u + i; }
    for (int i = 0; i < 4; ++i) q[i] = threadIdx.x * 0x0101010101ull + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 2) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
                if (KIND == 3) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(ua), "v"(b));
                if (KIND == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(ua), "v"(b), "v"(r[i]));
                if (KIND == 5) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[i]), "v"(r[(i + 1) & 7]));
                if (KIND == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
                if (KIND == 7) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(r[i]));
                if (KIND == 8) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 9) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 10) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(u[i]) : "v"(ua));
                if (KIND == 11) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
                if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ua) : "vcc");
                if (KIND == 13) asm volatile("v_bfe_u32 %0, %0, 3, 2" : "+v"(u[i]));
                if (KIND == 14) asm volatile("v_floor_f32 %0, %0" : "+v"(r[i]));
                if (KIND == 15) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(r[i]) : "v"(a), "v"(b));
                if (KIND == 16) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 17) asm volatile("v_qsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(q[i & 3]) : "v"(ua));
                if (KIND == 18) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 19) asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(u[i]));
                if (KIND == 20) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ua));
                if (KIND == 21) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ua));
                if (KIND == 22) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r[i]) : "v"(u[i]));
                if (KIND == 23) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                if (KIND == 24) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                if (KIND == 25) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
                if (KIND == 26) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 27) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(u[i]) : "v"(ua));
                if (KIND == 28) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 29) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(r[i]) : "v"(u[i]));
                if (KIND == 30) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 clamp" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (KIND == 31) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += r[i] + (float)u[i]; for (int i = 0; i < 4; ++i) s += (float)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run_valu(const char *name, float *out)
{
    for (int w : {1, 2, 4}) {
        int iters = 1000; dim3 grid(256 * w), block(256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_valu<KIND>, grid, block, 0, 0, out, 10, 1.0001f, 0.5f, 0x3c003c00u, 0x07060504u);
        hipEventRecord(e0); hipLaunchKernelGGL(k_valu<KIND>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f, 0x3c003c00u, 0x07060504u); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("A %-22s waves/SIMD=%d  %.2f cyc/wave-instr/SIMD\n", name, w, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * w));
    }
}

// ---------------------------------------------------------------- B: MFMA cost, alone and beside VALU
// SHAPE 0: f32_32x32x16_f16   1: f32_16x16x32_f16   2: f32_32x32x2_f32   3: f32_4x4x1_16B_f32   4: i32_32x32x32_i8
// NV independent v_fma_f32 per MFMA, two accumulators alternate.
template <int SHAPE, int NV>
__global__ void k_mfma(float *out, int iters, float a, float b)
{
    half8_t ha, hb; for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x % 7); hb[i] = (_Float16)(i); }
    f32x16 c0 = {}, c1 = {}; f32x4 d0 = {}, d1 = {}; i32x16 e0 = {}, e1 = {};
    i32x4 ia = {1, 2, 3, 4}, ib = {5, 6, 7, (int)threadIdx.x};
    float fa = threadIdx.x, fb = 0.5f;
    float r[8]; for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (SHAPE == 0) { if (rep & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0); }
            if (SHAPE == 1) { if (rep & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d1, 0, 0, 0); else d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d0, 0, 0, 0); }
            if (SHAPE == 2) { if (rep & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0); }
            if (SHAPE == 3) { if (rep & 1) d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, d1, 0, 0, 0); else d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, d0, 0, 0, 0); }
            if (SHAPE == 4) { if (rep & 1) e1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, e1, 0, 0, 0); else e0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, e0, 0, 0, 0); }
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i & 7]) : "v"(a), "v"(b));
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + (float)(e0[i] + e1[i]); for (int i = 0; i < 4; ++i) s += d0[i] + d1[i];
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE, int NV> void run_mfma(const char *name, float *out)
{
    for (int w : {1, 2}) {
        int iters = 500; dim3 grid(256 * w), block(256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k_mfma<SHAPE, NV>), grid, block, 0, 0, out, 10, 1.0001f, 0.5f);
        hipEventRecord(e0); hipLaunchKernelGGL((k_mfma<SHAPE, NV>), grid, block, 0, 0, out, iters, 1.0001f, 0.5f); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("B %-20s +%2d v_fma  waves/SIMD=%d  %.1f cyc per (MFMA + %d fma) per SIMD\n", name, NV, w, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * w), NV);
    }
}
template <int SHAPE> void run_mfma_all(const char *name, float *out)
{
    run_mfma<SHAPE, 0>(name, out); run_mfma<SHAPE, 4>(name, out); run_mfma<SHAPE, 8>(name, out);
    run_mfma<SHAPE, 12>(name, out); run_mfma<SHAPE, 16>(name, out); run_mfma<SHAPE, 24>(name, out);
}

// ---------------------------------------------------------------- C: lane maps of 32x32x16 f16
// Expected (cdna_hip_programming.md section 3): A[row l&31][k = 8(l>>5) + j], B[k = 8(l>>5) + j][col l&31],
// D reg i of lane l = D[row (i&3) + 8(i>>2) + 4(l>>5)][col l&31].
__global__ void k_layout(float *dout, float *d2out)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    half8_t a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        a[j] = (_Float16)(1 + r + 32 * k);                 // A[m][k] = 1 + m + 32 k   (<= 512, exact in f16)
        b[j] = (_Float16)((k == (r & 15)) ? 1.0f : 0.0f);  // B[k][n] = (k == n % 16)  -> D[m][n] = 1 + m + 32 (n % 16)
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) dout[l * 16 + i] = c[i];
    // accumulator tile X = c (32x32: X[m][n] = 1 + m + 32 (n % 16)) reused as the B operand: Y = T * X with
    // T[m'][k] = (k == (m' + 3) % 32) i.e. Y[m'][n] = X[(m' + 3) % 32][n]; k-step s uses registers 8s..8s+7,
    // whose element j of lane half h is row 16 s + 8 (j >> 2) + 4 h + (j & 3) of X.
    f32x16 y = {};
    for (int s = 0; s < 2; ++s) {
        half8_t xb, ta;
        for (int j = 0; j < 8; ++j) {
            xb[j] = (_Float16)c[8 * s + j];
            const int krow = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            ta[j] = (_Float16)((krow == ((r + 3) & 31)) ? 1.0f : 0.0f);
        }
        y = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta, xb, y, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) d2out[l * 16 + i] = y[i];
}

int main()
{
    float *out; hipMalloc(&out, 256 * 4 * 1024 * 64 * sizeof(float));
    {   // C first: cheap and decisive
        float *d, *d2; hipMalloc(&d, 64 * 16 * 4); hipMalloc(&d2, 64 * 16 * 4);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d, d2);
        std::vector<float> h(1024), h2(1024); hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), d2, 4096, hipMemcpyDeviceToHost);
        int bad = 0, bad2 = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), n = l & 31;
            if (h[l * 16 + i] != (float)(1 + m + 32 * (n % 16))) ++bad;
            if (h2[l * 16 + i] != (float)(1 + ((m + 3) & 31) + 32 * (n % 16))) ++bad2;
        }
        printf("C lane maps of v_mfma_f32_32x32x16_f16: %d of 1024 accumulator entries differ from the expected map; "
               "accumulator-as-B-operand: %d differ\n", bad, bad2);
    }
    run_mfma_all<0>("f32_32x32x16_f16", out); run_mfma_all<1>("f32_16x16x32_f16", out); run_mfma_all<2>("f32_32x32x2_f32", out);
    run_mfma_all<3>("f32_4x4x1_16B_f32", out); run_mfma_all<4>("i32_32x32x32_i8", out);
    run_valu<0>("v_fma_f32", out); run_valu<15>("v_fma_f32 clamp", out); run_valu<23>("v_sub_f32", out); run_valu<24>("v_max_f32", out);
    run_valu<6>("v_med3_f32", out); run_valu<14>("v_floor_f32", out);
    run_valu<1>("v_pk_fma_f16", out); run_valu<30>("v_pk_fma_f16 clamp", out); run_valu<2>("v_pk_add_f16", out); run_valu<25>("v_pk_max_f16", out);
    run_valu<3>("v_fma_mix_f32", out); run_valu<4>("v_fma_mixlo_f16", out); run_valu<5>("v_cvt_pkrtz_f16_f32", out); run_valu<22>("v_cvt_f32_f16", out);
    run_valu<7>("v_cvt_pk_u8_f32", out); run_valu<29>("v_cvt_f32_ubyte0", out);
    run_valu<8>("v_and_or_b32", out); run_valu<9>("v_or3_b32", out); run_valu<10>("v_alignbyte_b32", out); run_valu<26>("v_perm_b32", out);
    run_valu<27>("v_lshl_or_b32", out); run_valu<13>("v_bfe_u32", out); run_valu<19>("v_lshrrev_b32", out); run_valu<20>("v_and_b32", out);
    run_valu<21>("v_xor_b32", out); run_valu<31>("v_add_u32", out);
    run_valu<11>("v_cmp_gt_f32", out); run_valu<12>("v_cndmask_b32", out);
    run_valu<16>("v_sad_u8", out); run_valu<17>("v_qsad_pk_u16_u8", out); run_valu<18>("v_pk_mad_u16", out); run_valu<28>("v_dot4_u32_u8", out);
    return 0;
}
