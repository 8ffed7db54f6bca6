// Shader-clock probe: a single wave runs N dependent f64 FMAs; wall_clock64 (100 MHz) gives the time, the known issue
// latency gives the clock.  Run alone and while a second stream keeps all CUs busy.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/clock_probe tools/probes/clock_probe.hip && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_chain(double* out, int n, unsigned long long* ticks)
{
    double a = out[0], b = 1.0000001, c = 1e-9;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c);
        a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c);
    }
    const unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

__global__ void k_busy(double* out, int n)
{
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    for (int i = 0; i < n; i++) { a = __builtin_fma(a, b, c); b = __builtin_fma(b, a, c); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}

int main()
{
    double* d; unsigned long long* t; double* big;
    hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8); hipMalloc(&t, 8); hipMalloc(&big, 2048 * 256 * 8); hipMemset(big, 0, 2048 * 256 * 8);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int n = 20000;       // 160 k dependent FMAs
    for (int rep = 0; rep < 6; rep++) {
        const bool busy = rep >= 3;
        if (busy) hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s2, big, 400000);
        if (busy) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, s2); }
        unsigned long long h = 0;
        // a short kernel, then the same after 2 ms of back-to-back probes (does the clock ramp?)
        for (int k = 0; k < 3; k++) {
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, d, n, t);
            hipStreamSynchronize(s1);
            hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
            printf("%s probe %d: %.1f us for %d dependent f64 FMAs = %.2f ns each\n", busy ? "busy" : "idle", k, h / 100.0, 8 * n, h * 10.0 / (8.0 * n));
        }
        if (busy) hipStreamSynchronize(s2);
    }
    return 0;
}
