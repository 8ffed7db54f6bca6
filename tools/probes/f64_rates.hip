// Probe (GPU box): issue rate of v_mfma_f64_16x16x4 against v_fma_f64 on gfx950, one wave per SIMD and four waves per SIMD.
// build + run:  hipcc -O3 --offload-arch=gfx950 tools/probes/f64_rates.hip -o /tmp/f64_rates && /tmp/f64_rates
// Prints cycles per instruction per wave and FLOP per clock per CU.  Evidence for DESIGN.md: on MI355X the f64 matrix pipe has
// the vector pipe's FLOP rate, so moving the sparse 6x3 Schur products onto it (at < 60 % tile fill) cannot pay.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma(double* out, int iters, long long* cyc)
{
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void k_fma(double* out, int iters, long long* cyc)
{
    double a[8];
    for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 1e-3 + k;
    const double x = 1.0 + 1e-9, y = 1e-12;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = fma(a[k], x, y);
    }
    const long long t1 = clock64();
    double s = 0;
    for (int k = 0; k < 8; k++) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void k_fma_chain(double* out, int iters, long long* cyc)
{
    double a = threadIdx.x * 1e-3;
    const double x = 1.0 + 1e-9, y = 1e-12;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) { a = fma(a, x, y); a = fma(a, x, y); a = fma(a, x, y); a = fma(a, x, y); }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    double* out; long long* cyc; long long h;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    const int iters = 4096;
    for (int threads : {256, 1024}) {       // one workgroup on one CU: 1 or 4 waves per SIMD
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(threads), 0, 0, out, iters, cyc); hipDeviceSynchronize();
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(threads), 0, 0, out, iters, cyc); hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double per = (double)h / (4.0 * iters);
        printf("mfma_f64_16x16x4  %4d threads: %.1f clock64-cycles per MFMA per wave -> %.0f FLOP/clk/CU\n", threads, per, 2048.0 * (threads / 64) / per);
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(threads), 0, 0, out, iters, cyc); hipDeviceSynchronize();
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(threads), 0, 0, out, iters, cyc); hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double perf = (double)h / (8.0 * iters);
        printf("v_fma_f64         %4d threads: %.1f cycles per FMA per wave -> %.0f FLOP/clk/CU\n", threads, perf, 128.0 * (threads / 64) / perf);
    }
    hipLaunchKernelGGL(k_fma_chain, dim3(1), dim3(64), 0, 0, out, iters, cyc); hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("v_fma_f64 dependent chain: %.1f cycles per FMA\n", (double)h / (4.0 * iters));
    return 0;
}
