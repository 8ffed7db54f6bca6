// Probe (GPU box): what does a wave-instruction of k_fast_strips' mix cost on a gfx950 SIMD at 1, 2, 4 and 8 resident waves per SIMD?
// build + run:  hipcc -O3 --offload-arch=gfx950 tools/probes/fast_mix.hip -o /tmp/fast_mix && /tmp/fast_mix
// Every CU gets ONE workgroup of 256 x W threads (W waves per SIMD, resident together by construction: its LDS request fills the
// CU; W = 8 is two workgroups of 1024 threads), every wave runs the same unrolled stream of independent instructions and times itself with
// s_memtime.  Reported per mix and W:
//   wave   = cycles one wave needs per instruction of its own stream (latency seen by the wave)
//   simd   = cycles the SIMD spends per wave-instruction = wave / W   (the issue cost that bounds a kernel)
// Mixes: plain v_add_u32; v_perm_b32; the packed-u16 ops of stage 1 (v_pk_min/max/add/sub_u16 with clamp); v_min3/v_max3_i32 of
// the score tree; ds_read_b32; and stage 1's own proportions (5 ds_read_b32 : 10 v_perm_b32 : 24 v_pk_* : 4 other VALU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

enum { MIX_ADD = 0, MIX_PERM, MIX_PK, MIX_MINMAX3, MIX_LDS, MIX_STAGE1, MIX_ADD_E64, MIX_MIN_E32, MIX_COUNT };
static const char* kNames[MIX_COUNT] = {"v_add_u32", "v_perm_b32", "v_pk_*_u16", "v_min3/max3_i32", "ds_read_b32", "stage-1 mix", "v_add_u32_e64", "v_min_i32_e32"};
static const int kInstPerIter[MIX_COUNT] = {32, 32, 32, 32, 32, 43, 32, 32};

template <int MIX>
__global__ __launch_bounds__(1024) void k_probe(uint32_t* out, int iters, unsigned long long* cyc)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t k1 = 0x00010001u + blockIdx.x, k2 = 0x0C020C00u;
    const uint32_t addr = (threadIdx.x & 63) * 4;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = (unsigned long long)clock64();
    for (int i = 0; i < iters; i++) {
        if (MIX == MIX_ADD) {
            REP4(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1));)
        } else if (MIX == MIX_ADD_E64) {     // the same add in the 64-bit VOP3 encoding: is the cost the encoding's or the operation's?
            REP4(asm volatile("v_add_u32_e64 %0, %0, %8\n v_add_u32_e64 %1, %1, %8\n v_add_u32_e64 %2, %2, %8\n v_add_u32_e64 %3, %3, %8\n"
                              "v_add_u32_e64 %4, %4, %8\n v_add_u32_e64 %5, %5, %8\n v_add_u32_e64 %6, %6, %8\n v_add_u32_e64 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1));)
        } else if (MIX == MIX_MIN_E32) {
            REP4(asm volatile("v_min_i32_e32 %0, %0, %8\n v_max_i32_e32 %1, %1, %8\n v_min_i32_e32 %2, %2, %8\n v_max_i32_e32 %3, %3, %8\n"
                              "v_min_i32_e32 %4, %4, %8\n v_max_i32_e32 %5, %5, %8\n v_min_i32_e32 %6, %6, %8\n v_max_i32_e32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1));)
        } else if (MIX == MIX_PERM) {
            REP4(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                              "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2));)
        } else if (MIX == MIX_PK) {
            REP4(asm volatile("v_pk_min_u16 %0, %0, %8\n v_pk_max_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_sub_u16 %3, %3, %8 clamp\n"
                              "v_pk_min_u16 %4, %4, %8\n v_pk_max_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_sub_u16 %7, %7, %8 clamp\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1));)
        } else if (MIX == MIX_MINMAX3) {
            REP4(asm volatile("v_min3_i32 %0, %0, %8, %1\n v_max3_i32 %1, %1, %8, %2\n v_min3_i32 %2, %2, %8, %3\n v_max3_i32 %3, %3, %8, %4\n"
                              "v_min3_i32 %4, %4, %8, %5\n v_max3_i32 %5, %5, %8, %6\n v_min3_i32 %6, %6, %8, %7\n v_max3_i32 %7, %7, %8, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1));)
        } else if (MIX == MIX_LDS) {
            REP4(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                              "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr) : "memory");)
        } else {
            // one quad of stage 1: 5 aligned dword reads, 10 v_perm, 24 packed-u16 ops, 4 plain VALU (43 instructions)
            uint32_t r0, r1, r2, r3, r4;
            asm volatile("ds_read_b32 %0, %5\n ds_read_b32 %1, %5 offset:256\n ds_read_b32 %2, %5 offset:512\n ds_read_b32 %3, %5 offset:768\n ds_read_b32 %4, %5 offset:1024\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4) : "v"(addr) : "memory");
            asm volatile("v_perm_b32 %0, %0, %8, %13\n v_perm_b32 %1, %1, %8, %13\n v_perm_b32 %2, %2, %9, %13\n v_perm_b32 %3, %3, %9, %13\n"
                         "v_perm_b32 %4, %4, %10, %13\n v_perm_b32 %5, %5, %10, %13\n v_perm_b32 %6, %6, %11, %13\n v_perm_b32 %7, %7, %11, %13\n"
                         "v_perm_b32 %0, %0, %12, %13\n v_perm_b32 %1, %1, %12, %13\n"
                         "v_pk_sub_u16 %2, %2, %0 clamp\n v_pk_sub_u16 %3, %3, %1 clamp\n v_pk_add_u16 %4, %4, %0\n v_pk_add_u16 %5, %5, %1\n"
                         "v_pk_min_u16 %6, %6, %0\n v_pk_min_u16 %7, %7, %1\n v_pk_max_u16 %0, %0, %2\n v_pk_max_u16 %1, %1, %3\n"
                         "v_pk_min_u16 %2, %2, %4\n v_pk_min_u16 %3, %3, %5\n v_pk_max_u16 %4, %4, %6\n v_pk_max_u16 %5, %5, %7\n"
                         "v_pk_min_u16 %6, %6, %0\n v_pk_min_u16 %7, %7, %1\n v_pk_max_u16 %0, %0, %2\n v_pk_max_u16 %1, %1, %3\n"
                         "v_pk_sub_u16 %2, %2, %4 clamp\n v_pk_sub_u16 %3, %3, %5 clamp\n v_pk_sub_u16 %4, %4, %6 clamp\n v_pk_sub_u16 %5, %5, %7 clamp\n"
                         "v_pk_min_u16 %6, %6, %13\n v_pk_min_u16 %7, %7, %13\n v_pk_min_u16 %0, %0, %13\n v_pk_min_u16 %1, %1, %13\n"
                         "v_or_b32 %2, %2, %3\n v_or_b32 %4, %4, %5\n v_lshl_or_b32 %6, %7, 1, %6\n v_or_b32 %0, %0, %1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(k1));
        }
    }
    const unsigned long long t1 = (unsigned long long)clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    const unsigned long long w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[gridDim.x * (blockDim.x >> 6)] = t1 - t0; cyc[gridDim.x * (blockDim.x >> 6) + 1] = w1 - w0; }
}

template <int MIX>
static void run(int W, uint32_t* out, unsigned long long* cyc, int n_cu)
{
    const int iters = 2000;
    // W waves per SIMD, guaranteed resident together: ONE workgroup of 256 x min(W, 4) threads per CU (its LDS request fills the CU),
    // two such workgroups for W = 8
    const int wg_per_cu = W > 4 ? 2 : 1, threads = 256 * (W / wg_per_cu);
    const int grid = wg_per_cu * n_cu;
    const size_t lds = (160 * 1024 / wg_per_cu) - 2048;
    hipFuncSetAttribute((const void*)k_probe<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe<MIX>, dim3(grid), dim3(threads), lds, 0, out, iters, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    const size_t nw = (size_t)grid * (threads / 64);
    std::vector<unsigned long long> h(nw + 2);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    const double ghz = 0.1 * (double)h[nw] / (double)std::max<unsigned long long>(h[nw + 1], 1);    // wall_clock64 ticks at 100 MHz
    h.resize(nw);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2], n = (double)iters * kInstPerIter[MIX];
    // clock64() = s_memtime = shader cycles (MI355X_MICROARCH.md); the kernel's wall time gives the same figure in ns per SIMD
    printf("%-16s W=%d  cycles/instruction: wave %.2f  SIMD %.2f   (kernel %.3f ms -> %.3f ns per wave-instruction and SIMD; clock64 / wall_clock64 = %.2f GHz)\n", kNames[MIX], W, med / n, med / n / W,
           ms, 1e6 * ms / (n * W), ghz);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)8 * n_cu * 256 * 4); hipMalloc(&cyc, ((size_t)8 * n_cu * 4 + 2) * 8);
    printf("%s, %d CUs, clock %d kHz; one workgroup = 4 waves = one wave per SIMD; W workgroups per CU\n", p.name, n_cu, p.clockRate);
    for (int W : {1, 2, 4, 8}) {
        run<MIX_ADD>(W, out, cyc, n_cu); run<MIX_PERM>(W, out, cyc, n_cu); run<MIX_PK>(W, out, cyc, n_cu);
        run<MIX_MINMAX3>(W, out, cyc, n_cu); run<MIX_LDS>(W, out, cyc, n_cu); run<MIX_STAGE1>(W, out, cyc, n_cu);
        run<MIX_ADD_E64>(W, out, cyc, n_cu); run<MIX_MIN_E32>(W, out, cyc, n_cu);
    }
    return 0;
}
