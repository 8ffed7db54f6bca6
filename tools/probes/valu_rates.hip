// Probe (GPU box): issue cost of single VALU instructions on gfx950 at 4 waves per SIMD (one 1024-thread workgroup per CU, all CUs
// busy): kernel wall time / (wave-instructions per SIMD), in ns and in cycles of the clock measured by clock64 / wall_clock64.
// Which integer / byte / packed operations run at the 2-cycle rate of v_add_u32 / v_fma_f32, which at 4 cycles?
// build + run:  hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define OP2(name) name " %0, %0, %8\n " name " %1, %1, %8\n " name " %2, %2, %8\n " name " %3, %3, %8\n " name " %4, %4, %8\n " name " %5, %5, %8\n " name " %6, %6, %8\n " name " %7, %7, %8\n"
#define OP3(name) name " %0, %0, %8, %1\n " name " %1, %1, %8, %2\n " name " %2, %2, %8, %3\n " name " %3, %3, %8, %4\n " name " %4, %4, %8, %5\n " name " %5, %5, %8, %6\n " name " %6, %6, %8, %7\n " name " %7, %7, %8, %0\n"
#define OP3C(name) name " %0, %0, %8, 1\n " name " %1, %1, %8, 1\n " name " %2, %2, %8, 1\n " name " %3, %3, %8, 1\n " name " %4, %4, %8, 1\n " name " %5, %5, %8, 1\n " name " %6, %6, %8, 1\n " name " %7, %7, %8, 1\n"
#define OP1(name) name " %0, %8\n " name " %1, %8\n " name " %2, %8\n " name " %3, %8\n " name " %4, %8\n " name " %5, %8\n " name " %6, %8\n " name " %7, %8\n"
#define CMP(name) name " vcc, %0, %8\n " name " vcc, %1, %8\n " name " vcc, %2, %8\n " name " vcc, %3, %8\n " name " vcc, %4, %8\n " name " vcc, %5, %8\n " name " vcc, %6, %8\n " name " vcc, %7, %8\n"

#define KERNEL(id, body, clob)                                                                                          \
    __global__ __launch_bounds__(1024) void k_##id(uint32_t* out, int iters, unsigned long long* cyc)                   \
    {                                                                                                                   \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const uint32_t k1 = 0x00010001u + blockIdx.x;                                                                   \
        const unsigned long long w0 = wall_clock64(), t0 = clock64();                                                   \
        for (int i = 0; i < iters; i++) {                                                                               \
            REP4(asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1) : clob);) \
        }                                                                                                               \
        const unsigned long long t1 = clock64(), w1 = wall_clock64();                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }                                \
    }

KERNEL(add_u32, OP2("v_add_u32"), "memory")
KERNEL(sub_u32, OP2("v_sub_u32"), "memory")
KERNEL(and_b32, OP2("v_and_b32"), "memory")
KERNEL(or_b32, OP2("v_or_b32"), "memory")
KERNEL(xor_b32, OP2("v_xor_b32"), "memory")
KERNEL(lshlrev, OP2("v_lshlrev_b32"), "memory")
KERNEL(min_i32, OP2("v_min_i32"), "memory")
KERNEL(max_u32, OP2("v_max_u32"), "memory")
KERNEL(min_u16, OP2("v_min_u16"), "memory")
KERNEL(add_u16, OP2("v_add_u16"), "memory")
KERNEL(sub_u16, OP2("v_sub_u16"), "memory")
KERNEL(mul_u24, OP2("v_mul_u32_u24"), "memory")
KERNEL(add_f32, OP2("v_add_f32"), "memory")
KERNEL(min_f32, OP2("v_min_f32"), "memory")
KERNEL(max_f16, OP2("v_max_f16"), "memory")
KERNEL(pk_add_u16, OP2("v_pk_add_u16"), "memory")
KERNEL(pk_min_u16, OP2("v_pk_min_u16"), "memory")
KERNEL(pk_max_i16, OP2("v_pk_max_i16"), "memory")
KERNEL(pk_add_f16, OP2("v_pk_add_f16"), "memory")
KERNEL(pk_min_f16, OP2("v_pk_min_f16"), "memory")
KERNEL(pk_mul_lo_u16, OP2("v_pk_mul_lo_u16"), "memory")
KERNEL(cmp_gt_u32, CMP("v_cmp_gt_u32"), "vcc")
KERNEL(cndmask, OP2("v_cndmask_b32"), "vcc")
KERNEL(mov, OP1("v_mov_b32"), "memory")
KERNEL(fma_f32, OP3("v_fma_f32"), "memory")
KERNEL(mad_u24, OP3("v_mad_u32_u24"), "memory")
KERNEL(add3, OP3("v_add3_u32"), "memory")
KERNEL(and_or, OP3("v_and_or_b32"), "memory")
KERNEL(lshl_or, OP3C("v_lshl_or_b32"), "memory")
KERNEL(bfe, OP3C("v_bfe_u32"), "memory")
KERNEL(perm, OP3("v_perm_b32"), "memory")
KERNEL(alignbyte, OP3C("v_alignbyte_b32"), "memory")
KERNEL(min3_i32, OP3("v_min3_i32"), "memory")
KERNEL(max3_u32, OP3("v_max3_u32"), "memory")
KERNEL(med3_i32, OP3("v_med3_i32"), "memory")
KERNEL(min3_f32, OP3("v_min3_f32"), "memory")
KERNEL(sad_u8, OP3("v_sad_u8"), "memory")
KERNEL(msad_u8, OP3("v_msad_u8"), "memory")
KERNEL(sad_u16, OP3("v_sad_u16"), "memory")
KERNEL(sad_u32, OP3("v_sad_u32"), "memory")
KERNEL(dot4_u32_u8, OP3("v_dot4_u32_u8"), "memory")
KERNEL(dot2_u32_u16, OP3("v_dot2_u32_u16"), "memory")
KERNEL(pk_mad_u16, OP3("v_pk_mad_u16"), "memory")
KERNEL(cvt_pk_u8_f32, OP3("v_cvt_pk_u8_f32"), "memory")
KERNEL(bfi, OP3("v_bfi_b32"), "memory")
KERNEL(xad, OP3("v_xad_u32"), "memory")
KERNEL(lerp_u8, OP3("v_lerp_u8"), "memory")

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount, iters = 2000, W = 4;
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)n_cu * 1024 * 4); hipMalloc(&cyc, 16);
    printf("%d CUs; one 1024-thread workgroup per CU = %d waves per SIMD; %d x 32 instructions per wave\n", n_cu, W, iters);
#define RUN(id)                                                                                                            \
    {                                                                                                                      \
        hipFuncSetAttribute((const void*)k_##id, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                  \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0.f;                                       \
        for (int rep = 0; rep < 2; rep++) {                                                                                \
            hipEventRecord(e0, 0);                                                                                         \
            hipLaunchKernelGGL(k_##id, dim3(n_cu), dim3(1024), 150 * 1024, 0, out, iters, cyc);                            \
            hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);                               \
        }                                                                                                                  \
        unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);                                             \
        const double ghz = 0.1 * (double)h[0] / (double)h[1], ns = 1e6 * (ms - 0.008) / ((double)iters * 32 * W);           \
        printf("%-18s %.3f ns per wave-instruction and SIMD = %.2f cycles at %.2f GHz\n", #id, ns, ns * ghz, ghz);          \
    }
    RUN(add_u32) RUN(sub_u32) RUN(and_b32) RUN(or_b32) RUN(xor_b32) RUN(lshlrev) RUN(min_i32) RUN(max_u32) RUN(min_u16) RUN(add_u16) RUN(sub_u16)
    RUN(mul_u24) RUN(add_f32) RUN(min_f32) RUN(max_f16) RUN(pk_add_u16) RUN(pk_min_u16) RUN(pk_max_i16) RUN(pk_add_f16) RUN(pk_min_f16) RUN(pk_mul_lo_u16)
    RUN(cmp_gt_u32) RUN(cndmask) RUN(mov) RUN(fma_f32) RUN(mad_u24) RUN(add3) RUN(and_or) RUN(lshl_or) RUN(bfe) RUN(perm) RUN(alignbyte)
    RUN(min3_i32) RUN(max3_u32) RUN(med3_i32) RUN(min3_f32) RUN(sad_u8) RUN(msad_u8) RUN(sad_u16) RUN(sad_u32) RUN(dot4_u32_u8) RUN(dot2_u32_u16)
    RUN(pk_mad_u16) RUN(cvt_pk_u8_f32) RUN(bfi) RUN(xad) RUN(lerp_u8)
    return 0;
}
