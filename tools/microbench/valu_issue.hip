// valu_issue.hip -- what one wave64 VALU instruction costs a gfx950 SIMD, by class (the roofline's "peak" assumed 2 cycles
// for 32-bit and 4 for 64-bit instructions without a measurement).  Every kernel issues long runs of INDEPENDENT instructions of one
// kind (8 accumulators per lane, no memory traffic) from 8 waves per SIMD on every SIMD of the chip; the shader clock during the
// kernel comes from s_memtime (shader cycles) against s_memrealtime (100 MHz).  Output: one JSON line per instruction kind with
// Ginst/s (wave-instructions) and cycles per instruction per SIMD at the clock the chip held.
//   hipcc --offload-arch=gfx950 -O2 -o build_ab/valu_issue tools/microbench/valu_issue.hip && build_ab/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr int kIter = 4096, kUnroll = 32;        // instructions per lane = kIter * kUnroll

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define KERNEL(NAME, TYPE, INIT, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(TYPE *out, unsigned long long *clk)                          \
    {                                                                                                        \
        TYPE a[8];                                                                                           \
        for (int q = 0; q < 8; q++) a[q] = (TYPE)(INIT + q + threadIdx.x);                                   \
        TYPE c = (TYPE)(INIT + 1), d = (TYPE)(INIT + 3);                                                     \
        unsigned long long t0, r0, t1, r1;                                                                   \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));       \
        for (int it = 0; it < kIter; it++) {                                                                 \
            REP32(ASM)                                                                                       \
        }                                                                                                    \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));       \
        TYPE s = a[0];                                                                                       \
        for (int q = 1; q < 8; q++) s += a[q];                                                               \
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;                                              \
        if (threadIdx.x == 0) { clk[4 * blockIdx.x] = t1 - t0; clk[4 * blockIdx.x + 1] = r1 - r0; clk[4 * blockIdx.x + 2] = r0; clk[4 * blockIdx.x + 3] = r1; } \
    }

#define OP_ADD_U32(q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[q]) : "v"(c));
#define OP_AND_B32(q) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[q]) : "v"(c));
#define OP_ADD_F32(q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[q]) : "v"(c));
#define OP_FMA_F32(q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(c), "v"(d));
#define OP_CNDMASK(q) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q]) : "v"(c));      // (vcc is whatever it is: not declared clobbered, or the compiler pads every use with s_nop)
#define OP_CVT_F64_F32(q) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(a[q]) : "v"(cf));
#define OP_ADD_F64(q) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[q]) : "v"(c));
#define OP_MUL_F64(q) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[q]) : "v"(c));
#define OP_FMA_F64(q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[q]) : "v"(c), "v"(d));
#define OP_CMP_F64(q) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[q]), "v"(c) : "vcc");
#define OP_CMP_U32(q) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[q]), "v"(c) : "vcc");
#define OP_MOV_B32(q) asm volatile("v_mov_b32 %0, %1" : "+v"(a[q]) : "v"(c));
#define OP_LSHL_B64(q) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[q]));

KERNEL(k_add_u32, unsigned, 1u, OP_ADD_U32)
KERNEL(k_and_b32, unsigned, 0xffffu, OP_AND_B32)
KERNEL(k_add_f32, float, 1.0f, OP_ADD_F32)
KERNEL(k_fma_f32, float, 1.0f, OP_FMA_F32)
KERNEL(k_cndmask_b32, unsigned, 1u, OP_CNDMASK)
KERNEL(k_mov_b32, unsigned, 1u, OP_MOV_B32)
KERNEL(k_cmp_u32, unsigned, 1u, OP_CMP_U32)
KERNEL(k_add_f64, double, 1.0, OP_ADD_F64)
KERNEL(k_mul_f64, double, 1.0, OP_MUL_F64)
KERNEL(k_fma_f64, double, 1.0, OP_FMA_F64)
KERNEL(k_cmp_f64, double, 1.0, OP_CMP_F64)
KERNEL(k_lshl_b64, unsigned long long, 1ull, OP_LSHL_B64)

// v_cndmask in context: the mask in an SGPR pair instead of vcc; alternating with plain adds; behind the compare that makes its mask
#define OP_CND_SGPR(q) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[q]) : "v"(c), "s"(msk));
#define OP_CND_ALT(q) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %0, %0, %1" : "+v"(a[q]) : "v"(c));
#define OP_CND_ALT3(q) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %0, %0, %1\n\tv_and_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(a[q]) : "v"(c));
#define OP_CMP_CND(q) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q]) : "v"(c) : "vcc");
#define OP_CMP64_CND(q) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc" : : "v"(a[q]), "v"(c), "v"(bsel), "v"(ci) : "vcc");
__global__ __launch_bounds__(256) void k_cnd_sgpr(unsigned *out, unsigned long long *clk)
{
    unsigned a[8]; for (int q = 0; q < 8; q++) a[q] = 1u + q + threadIdx.x;
    unsigned c = 2u; unsigned long long msk = 0x5555555555555555ull + blockIdx.x;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int it = 0; it < kIter; it++) { REP32(OP_CND_SGPR) }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    unsigned s = 0; for (int q = 0; q < 8; q++) s += a[q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[4 * blockIdx.x] = t1 - t0; clk[4 * blockIdx.x + 1] = r1 - r0; clk[4 * blockIdx.x + 2] = r0; clk[4 * blockIdx.x + 3] = r1; }
}
KERNEL(k_cnd_alt, unsigned, 1u, OP_CND_ALT)
KERNEL(k_cnd_alt3, unsigned, 1u, OP_CND_ALT3)
KERNEL(k_cmp_cnd, unsigned, 1u, OP_CMP_CND)
__global__ __launch_bounds__(256) void k_cmp64_cnd(double *out, unsigned long long *clk)
{
    double a[8]; for (int q = 0; q < 8; q++) a[q] = 1.0 + q + threadIdx.x;
    double c = 5.0; unsigned bsel = threadIdx.x, ci = 7;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int it = 0; it < kIter; it++) { REP32(OP_CMP64_CND) }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    double s = bsel; for (int q = 0; q < 8; q++) s += a[q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[4 * blockIdx.x] = t1 - t0; clk[4 * blockIdx.x + 1] = r1 - r0; clk[4 * blockIdx.x + 2] = r0; clk[4 * blockIdx.x + 3] = r1; }
}

// a mix like the fused loop's: 3 of 4 instructions of the 64-bit classes, every instruction independent of its 7 predecessors
#define OP_MFMA(q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[q]) : "v"(c), "v"(d));
#define OP_MADD(q) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[q]) : "v"(c));
#define OP_MMUL(q) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[q]) : "v"(d));
#define OP_MU32(q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[q]) : "v"(ci));
__global__ __launch_bounds__(256) void k_mix_3of4_f64(double *out, unsigned long long *clk)
{
    double a[8]; unsigned b[8];
    for (int q = 0; q < 8; q++) { a[q] = 1.0 + q + threadIdx.x; b[q] = q + threadIdx.x; }
    double c = 1.000001, d = 0.999999; unsigned ci = 3;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int it = 0; it < kIter; it++) { REP8(OP_MFMA) REP8(OP_MU32) REP8(OP_MADD) REP8(OP_MMUL) }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    double s = 0; for (int q = 0; q < 8; q++) s += a[q] + b[q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[4 * blockIdx.x] = t1 - t0; clk[4 * blockIdx.x + 1] = r1 - r0; clk[4 * blockIdx.x + 2] = r0; clk[4 * blockIdx.x + 3] = r1; }
}

// the same 3 : 1 mix as ONE wave sees it in the fused loop: each instruction depends on the one before it (a chain)
#define OP_CHAIN(q) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_add_f64 %0, %0, %2\n\tv_mul_f64 %0, %0, %3\n\tv_add_u32 %1, %1, %4" \
                                 : "+v"(a[q]), "+v"(b[q]) : "v"(c), "v"(d), "v"(ci));
__global__ __launch_bounds__(256) void k_mix_chain(double *out, unsigned long long *clk)
{
    double a[8]; unsigned b[8];
    for (int q = 0; q < 8; q++) { a[q] = 1.0 + q + threadIdx.x; b[q] = q + threadIdx.x; }
    double c = 1.000001, d = 0.999999; unsigned ci = 3;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int it = 0; it < kIter; it++) { REP8(OP_CHAIN) }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    double s = 0; for (int q = 0; q < 8; q++) s += a[q] + b[q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[4 * blockIdx.x] = t1 - t0; clk[4 * blockIdx.x + 1] = r1 - r0; clk[4 * blockIdx.x + 2] = r0; clk[4 * blockIdx.x + 3] = r1; }
}

template <typename T, typename K>
static void run(const char *name, K kern, int width_bits, double insts_per_lane, int wg_per_cu = 8)
{
    int dev = 0; hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, dev));
    const int simds = pr.multiProcessorCount * 4, blocks = pr.multiProcessorCount * wg_per_cu;    // workgroups of 4 waves: wg_per_cu waves per SIMD
    T *out; unsigned long long *clk;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(T))); CHK(hipMalloc(&clk, (size_t)blocks * 32));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, clk);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long *h = (unsigned long long *)malloc((size_t)blocks * 32);
    CHK(hipMemcpy(h, clk, (size_t)blocks * 32, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    unsigned long long first = ~0ull, last_start = 0, last_end = 0;
    for (int b = 0; b < blocks; b++) {
        cyc += (double)h[4 * b]; real += (double)h[4 * b + 1];
        if (h[4 * b + 2] < first) first = h[4 * b + 2];
        if (h[4 * b + 2] > last_start) last_start = h[4 * b + 2];
        if (h[4 * b + 3] > last_end) last_end = h[4 * b + 3];
    }
    const double ghz = cyc / real * 0.1;                                      // s_memrealtime ticks at 100 MHz
    const double waves = (double)blocks * 4, winst = waves * insts_per_lane;  // wave-instructions per launch
    const double ginst = winst * reps / (ms * 1e-3) / 1e9;
    // cycles a SIMD spends per wave-instruction: from the kernel's own span on the chip (first start .. last end of the last launch,
    // 100-MHz ticks -> shader cycles at the clock measured in the same kernel), launch gaps excluded
    const double span_cyc = (double)(last_end - first) * 10.0 * ghz;
    const double cyc_per_inst = span_cyc * simds / winst;
    printf("{\"inst\": \"%s\", \"width_bits\": %d, \"waves_per_simd\": %d, \"Ginst_per_s\": %.1f, \"sclk_GHz\": %.3f, "
           "\"cycles_per_wave_inst_per_simd\": %.3f, \"Ginst_per_s_at_2p4GHz\": %.1f, \"start_spread_over_span\": %.3f}\n",
           name, width_bits, wg_per_cu, ginst, ghz, cyc_per_inst, simds * 2.4 / cyc_per_inst, (double)(last_start - first) / (double)(last_end - first));
    fflush(stdout);
    free(h); CHK(hipFree(out)); CHK(hipFree(clk));
}

int main()
{
    const double n = (double)kIter * kUnroll;
    run<unsigned>("v_add_u32", k_add_u32, 32, n);
    run<unsigned>("v_and_b32", k_and_b32, 32, n);
    run<unsigned>("v_mov_b32", k_mov_b32, 32, n);
    run<unsigned>("v_cndmask_b32", k_cndmask_b32, 32, n);
    run<unsigned>("v_cndmask_b32_e64 (mask in an SGPR pair)", k_cnd_sgpr, 32, n);
    run<unsigned>("v_cndmask_b32 vcc ; v_add_u32 (1:1)", k_cnd_alt, 32, 2 * n);
    run<unsigned>("v_cndmask_b32 vcc ; 3 x 32-bit ALU (1:3)", k_cnd_alt3, 32, 4 * n);
    run<unsigned>("v_cmp_lt_u32 vcc ; v_cndmask_b32 vcc", k_cmp_cnd, 32, 2 * n);
    run<double>("v_cmp_lt_f64 vcc ; v_cndmask_b32 vcc", k_cmp64_cnd, 0, 2 * n);
    run<unsigned>("v_cmp_lt_u32", k_cmp_u32, 32, n);
    run<float>("v_add_f32", k_add_f32, 32, n);
    run<float>("v_fma_f32", k_fma_f32, 32, n);
    run<double>("v_add_f64", k_add_f64, 64, n);
    run<double>("v_mul_f64", k_mul_f64, 64, n);
    run<double>("v_fma_f64", k_fma_f64, 64, n);
    run<double>("v_cmp_lt_f64", k_cmp_f64, 64, n);
    run<unsigned long long>("v_lshlrev_b64", k_lshl_b64, 64, n);
    run<double>("mix 3:1 (fma_f64, add_u32, add_f64, mul_f64), independent", k_mix_3of4_f64, 0, n);
    run<double>("mix 3:1, each instruction dependent on its predecessor", k_mix_chain, 0, n);
    // the fused loop runs 7 waves per SIMD; how the dependent mix degrades with fewer waves to hide its latency
    run<double>("mix 3:1 dependent", k_mix_chain, 0, n, 4);
    run<double>("mix 3:1 dependent", k_mix_chain, 0, n, 2);
    run<double>("mix 3:1 dependent", k_mix_chain, 0, n, 1);
    run<double>("v_fma_f64", k_fma_f64, 64, n, 4);
    run<double>("v_fma_f64", k_fma_f64, 64, n, 1);
    return 0;
}
