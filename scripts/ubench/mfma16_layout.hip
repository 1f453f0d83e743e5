// Lane maps of v_mfma_f32_16x16x32_f16 on gfx950, checked with exact small integers, and the code the compiler
// emits for the f16 hi/lo split of an f32 accumulator (printed by `make mfma16_layout.s`).
//   hypothesis 1: A[m = l%16][k = 8(l/16) + j], B[k = 8(l/16) + j][n = l%16]           (j = 0..7 in the lane's 8 halves)
//   hypothesis 2: k = 4(l/16) + (j&3) + 16(j>>2)                                        (two stacked K=16 halves)
//   D reg r of lane l = D[m = 4(l/16) + r][n = l%16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_probe(const float *A, const float *B, float *D, int hyp)
{
    const int l = threadIdx.x, r16 = l & 15, q = l >> 4;
    half8_t a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = hyp == 1 ? 8 * q + j : 4 * q + (j & 3) + 16 * (j >> 2);
        a[j] = (_Float16)A[r16 * 32 + k];
        b[j] = (_Float16)B[k * 16 + r16];
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + r16] = c[r];
}

// the split used by the threshold kernel: hi = rtz f16 of x, lo = f16 of (x - hi)
__global__ void k_split(const float *x, unsigned *hi, unsigned *lo)
{
    const int i = threadIdx.x;
    const float a = x[2 * i], b = x[2 * i + 1];
    auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const float ra = __builtin_fmaf((float)h[0], -1.0f, a), rb = __builtin_fmaf((float)h[1], -1.0f, b);
    auto g = __builtin_amdgcn_cvt_pkrtz(ra, rb);
    hi[i] = *reinterpret_cast<unsigned *>(&h);
    lo[i] = *reinterpret_cast<unsigned *>(&g);
}

int main()
{
    std::vector<float> A(16 * 32), B(32 * 16), D(256), ref(256);
    srand(1);
    for (auto &v : A) v = (float)(rand() % 9 - 4);
    for (auto &v : B) v = (float)(rand() % 9 - 4);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += A[m * 32 + k] * B[k * 16 + n]; ref[m * 16 + n] = s; }
    float *dA, *dB, *dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    for (int hyp : {1, 2}) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 256; ++i) bad += D[i] != ref[i];
        printf("v_mfma_f32_16x16x32_f16 hypothesis %d: %d of 256 entries differ\n", hyp, bad);
    }
    std::vector<float> x(128); for (int i = 0; i < 128; ++i) x[i] = 255.0f * rand() / RAND_MAX;
    float *dx; unsigned *dh, *dl; hipMalloc(&dx, 512); hipMalloc(&dh, 256); hipMalloc(&dl, 256);
    hipMemcpy(dx, x.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dx, dh, dl);
    std::vector<unsigned> h(64), lo(64); hipMemcpy(h.data(), dh, 256, hipMemcpyDeviceToHost); hipMemcpy(lo.data(), dl, 256, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 2; ++e) {
        _Float16 hh, ll; unsigned short hb = (h[i] >> (16 * e)) & 0xFFFF, lb = (lo[i] >> (16 * e)) & 0xFFFF;
        __builtin_memcpy(&hh, &hb, 2); __builtin_memcpy(&ll, &lb, 2);
        double err = (double)x[2 * i + e] - ((double)(float)hh + (double)(float)ll); if (err < 0) err = -err; if (err > worst) worst = err;
    }
    printf("f16 hi/lo split of values in [0,255]: worst |x - (hi + lo)| = %.3g\n", worst);
    return 0;
}
