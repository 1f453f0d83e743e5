// FETCH_SIZE / WRITE_SIZE calibration for the threshold kernel's access shape: every lane reads
// one dword (4 B) of a row, lanes consecutive (256 B per wave-instruction), and writes one dword.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy_dword(const unsigned *__restrict__ in, unsigned *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i] + 1u;
}
int main()
{
    const size_t bytes = 580ull << 20, n = bytes / 4;   // 580 MiB each way, >> 256 MiB L3
    unsigned *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(copy_dword, dim3(8192), dim3(256), 0, 0, a, b, n);
    hipDeviceSynchronize();
    printf("copy_dword: %zu bytes read, %zu bytes written per launch\n", bytes, bytes);
    return 0;
}
