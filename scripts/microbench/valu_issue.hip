// Microbenchmark: ISSUE cost (cycles per wave64 instruction per SIMD) of the vector instruction classes the solver
// kernels are made of, with 1, 2 and 3 wavefronts per SIMD.  Used to price the VALU issue floor of bench.py's roofline
// with measured figures instead of "FP64 = 4 cycles, everything else = 2" (profiles/r03_valu_issue_costs.json).
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
// Every class runs as 8 independent streams (no instruction depends on the one before it) of inline assembly, 256
// instructions per loop iteration; cycles = kernel time x the clock measured in-kernel (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// OP(i): one instruction of the class on register set i.  d[i] are doubles (v[..:..]), u[i] 32-bit, dd are outputs.
#define DEFINE_KERNEL(NAME, ASM_LINE)                                                                                   \
    __global__ void __launch_bounds__(64) NAME(double *out, int iters, unsigned long long *clk, int active)           \
    {                                                                                                                   \
        if ((int)threadIdx.x >= active) return;   /* sparse-EXEC runs: only lanes [0, active) execute */               \
        double d0 = 1.0 + threadIdx.x * 1e-9, d1 = 1.1, d2 = 1.2, d3 = 1.3, d4 = 1.4, d5 = 1.5, d6 = 1.6, d7 = 1.7;     \
        double e = 1.000000001 + blockIdx.x * 1e-12, f = 1e-9;                                                          \
        unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3, u4 = 4, u5 = 5, u6 = 6, u7 = 7;                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();             \
        for (int it = 0; it < iters; ++it) {                                                                            \
            _Pragma("unroll") for (int k = 0; k < 32; ++k) {                                                            \
                asm volatile(ASM_LINE(0) ASM_LINE(1) ASM_LINE(2) ASM_LINE(3) ASM_LINE(4) ASM_LINE(5) ASM_LINE(6) ASM_LINE(7) \
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7),          \
                               "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)            \
                             : "v"(e), "v"(f)                                                                           \
                             : "vcc", "s10", "s11", "s12", "s13");                                                                                \
            }                                                                                                           \
        }                                                                                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();             \
        out[(size_t)blockIdx.x * 64 + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + (double)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7); \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                                \
    }

// operand numbering of the asm: %0..%7 = d0..d7, %8..%15 = u0..u7, %16 = e, %17 = f
#define L_FMA(i) "v_fma_f64 %" #i ", %" #i ", %16, %17\n"
#define L_MUL(i) "v_mul_f64 %" #i ", %" #i ", %16\n"
#define L_ADD(i) "v_add_f64 %" #i ", %" #i ", %17\n"
#define L_RCP(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define L_RSQ(i) "v_rsq_f64 %" #i ", %" #i "\n"
#define L_MAX(i) "v_max_f64 %" #i ", %" #i ", %16\n"
#define L_CMP(i) "v_cmp_lt_f64 vcc, %" #i ", %16\n"
#define L_FIXUP(i) "v_div_fixup_f64 %" #i ", %" #i ", %16, %17\n"
#define L_LDEXP(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define L_RNDNE(i) "v_rndne_f64 %" #i ", %" #i "\n"
#define L_CLASS(i) "v_cmp_class_f64 vcc, %" #i ", 3\n"
#define L_MOV64(i) "v_mov_b64 %" #i ", %16\n"
// 32-bit classes use u registers: operands %8..%15
#define L_MOV32(i) L_MOV32_##i
#define L_MOV32_0 "v_mov_b32 %8, %9\n"
#define L_MOV32_1 "v_mov_b32 %9, %10\n"
#define L_MOV32_2 "v_mov_b32 %10, %11\n"
#define L_MOV32_3 "v_mov_b32 %11, %12\n"
#define L_MOV32_4 "v_mov_b32 %12, %13\n"
#define L_MOV32_5 "v_mov_b32 %13, %14\n"
#define L_MOV32_6 "v_mov_b32 %14, %15\n"
#define L_MOV32_7 "v_mov_b32 %15, %8\n"
#define L_CND(i) L_CND_##i
#define L_CND_0 "v_cndmask_b32 %8, %8, %9, vcc\n"
#define L_CND_1 "v_cndmask_b32 %9, %9, %10, vcc\n"
#define L_CND_2 "v_cndmask_b32 %10, %10, %11, vcc\n"
#define L_CND_3 "v_cndmask_b32 %11, %11, %12, vcc\n"
#define L_CND_4 "v_cndmask_b32 %12, %12, %13, vcc\n"
#define L_CND_5 "v_cndmask_b32 %13, %13, %14, vcc\n"
#define L_CND_6 "v_cndmask_b32 %14, %14, %15, vcc\n"
#define L_CND_7 "v_cndmask_b32 %15, %15, %8, vcc\n"
#define L_ADDU(i) L_ADDU_##i
#define L_ADDU_0 "v_add_u32 %8, %8, %9\n"
#define L_ADDU_1 "v_add_u32 %9, %9, %10\n"
#define L_ADDU_2 "v_add_u32 %10, %10, %11\n"
#define L_ADDU_3 "v_add_u32 %11, %11, %12\n"
#define L_ADDU_4 "v_add_u32 %12, %12, %13\n"
#define L_ADDU_5 "v_add_u32 %13, %13, %14\n"
#define L_ADDU_6 "v_add_u32 %14, %14, %15\n"
#define L_ADDU_7 "v_add_u32 %15, %15, %8\n"
#define L_XOR(i) L_XOR_##i
#define L_XOR_0 "v_xor_b32 %8, %8, %9\n"
#define L_XOR_1 "v_xor_b32 %9, %9, %10\n"
#define L_XOR_2 "v_xor_b32 %10, %10, %11\n"
#define L_XOR_3 "v_xor_b32 %11, %11, %12\n"
#define L_XOR_4 "v_xor_b32 %12, %12, %13\n"
#define L_XOR_5 "v_xor_b32 %13, %13, %14\n"
#define L_XOR_6 "v_xor_b32 %14, %14, %15\n"
#define L_XOR_7 "v_xor_b32 %15, %15, %8\n"
#define L_CMPU(i) L_CMPU_##i
#define L_CMPU_0 "v_cmp_eq_u32 vcc, %8, %9\n"
#define L_CMPU_1 "v_cmp_eq_u32 vcc, %9, %10\n"
#define L_CMPU_2 "v_cmp_eq_u32 vcc, %10, %11\n"
#define L_CMPU_3 "v_cmp_eq_u32 vcc, %11, %12\n"
#define L_CMPU_4 "v_cmp_eq_u32 vcc, %12, %13\n"
#define L_CMPU_5 "v_cmp_eq_u32 vcc, %13, %14\n"
#define L_CMPU_6 "v_cmp_eq_u32 vcc, %14, %15\n"
#define L_CMPU_7 "v_cmp_eq_u32 vcc, %15, %8\n"
#define L_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 3, %16\n"
// a select of a double as the compiler emits it: one compare into VCC, two v_cndmask_b32 (3 instructions per line)
#define L_SEL(i) L_SEL_##i
#define L_SEL_0 "v_cmp_lt_f64 vcc, %0, %16\nv_cndmask_b32 %8, %8, %9, vcc\nv_cndmask_b32 %9, %9, %10, vcc\n"
#define L_SEL_1 "v_cmp_lt_f64 vcc, %1, %16\nv_cndmask_b32 %10, %10, %11, vcc\nv_cndmask_b32 %11, %11, %12, vcc\n"
#define L_SEL_2 "v_cmp_lt_f64 vcc, %2, %16\nv_cndmask_b32 %12, %12, %13, vcc\nv_cndmask_b32 %13, %13, %14, vcc\n"
#define L_SEL_3 "v_cmp_lt_f64 vcc, %3, %16\nv_cndmask_b32 %14, %14, %15, vcc\nv_cndmask_b32 %15, %15, %8, vcc\n"
#define L_SEL_4 L_SEL_0
#define L_SEL_5 L_SEL_1
#define L_SEL_6 L_SEL_2
#define L_SEL_7 L_SEL_3
// the same select with the mask in a scalar register pair (VOP3 form)
#define L_SELS(i) L_SELS_##i
#define L_SELS_0 "v_cmp_lt_f64 s[10:11], %0, %16\nv_cndmask_b32 %8, %8, %9, s[10:11]\nv_cndmask_b32 %9, %9, %10, s[10:11]\n"
#define L_SELS_1 "v_cmp_lt_f64 s[12:13], %1, %16\nv_cndmask_b32 %10, %10, %11, s[12:13]\nv_cndmask_b32 %11, %11, %12, s[12:13]\n"
#define L_SELS_2 "v_cmp_lt_f64 s[10:11], %2, %16\nv_cndmask_b32 %12, %12, %13, s[10:11]\nv_cndmask_b32 %13, %13, %14, s[10:11]\n"
#define L_SELS_3 "v_cmp_lt_f64 s[12:13], %3, %16\nv_cndmask_b32 %14, %14, %15, s[12:13]\nv_cndmask_b32 %15, %15, %8, s[12:13]\n"
#define L_SELS_4 L_SELS_0
#define L_SELS_5 L_SELS_1
#define L_SELS_6 L_SELS_2
#define L_SELS_7 L_SELS_3

// quarter-rate 32-bit integer multiply and the 64-bit multiply-add of address arithmetic
#define L_MULLO(i) L_MULLO_##i
#define L_MULLO_0 "v_mul_lo_u32 %8, %8, %9\n"
#define L_MULLO_1 "v_mul_lo_u32 %9, %9, %10\n"
#define L_MULLO_2 "v_mul_lo_u32 %10, %10, %11\n"
#define L_MULLO_3 "v_mul_lo_u32 %11, %11, %12\n"
#define L_MULLO_4 "v_mul_lo_u32 %12, %12, %13\n"
#define L_MULLO_5 "v_mul_lo_u32 %13, %13, %14\n"
#define L_MULLO_6 "v_mul_lo_u32 %14, %14, %15\n"
#define L_MULLO_7 "v_mul_lo_u32 %15, %15, %8\n"
// MIXED streams (2 instructions per line): does a 32-bit instruction issue in the shadow of an f64 one of the same wave /
// of the other waves of the SIMD, or do the costs add?
#define L_MIXADDU(i) L_FMA(i) L_ADDU(i)
#define L_MIXCND(i) L_FMA(i) L_CND(i)
#define L_MIXMOV64(i) L_FMA(i) L_MOV64X(i)
#define L_MOV64X(i) L_MOV64X_##i
#define L_MOV64X_0 "v_mov_b64 %1, %16\n"
#define L_MOV64X_1 "v_mov_b64 %2, %16\n"
#define L_MOV64X_2 "v_mov_b64 %3, %16\n"
#define L_MOV64X_3 "v_mov_b64 %4, %16\n"
#define L_MOV64X_4 "v_mov_b64 %5, %16\n"
#define L_MOV64X_5 "v_mov_b64 %6, %16\n"
#define L_MOV64X_6 "v_mov_b64 %7, %16\n"
#define L_MOV64X_7 "v_mov_b64 %0, %16\n"
#define L_MIXMULLO(i) L_FMA(i) L_MULLO(i)
#define L_MIXCMP(i) L_FMA(i) L_CMP(i)

DEFINE_KERNEL(k_fma, L_FMA)
DEFINE_KERNEL(k_mul, L_MUL)
DEFINE_KERNEL(k_add, L_ADD)
DEFINE_KERNEL(k_rcp, L_RCP)
DEFINE_KERNEL(k_rsq, L_RSQ)
DEFINE_KERNEL(k_max, L_MAX)
DEFINE_KERNEL(k_cmp, L_CMP)
DEFINE_KERNEL(k_fixup, L_FIXUP)
DEFINE_KERNEL(k_ldexp, L_LDEXP)
DEFINE_KERNEL(k_rndne, L_RNDNE)
DEFINE_KERNEL(k_class, L_CLASS)
DEFINE_KERNEL(k_mov64, L_MOV64)
DEFINE_KERNEL(k_mov32, L_MOV32)
DEFINE_KERNEL(k_cnd, L_CND)
DEFINE_KERNEL(k_addu, L_ADDU)
DEFINE_KERNEL(k_xor, L_XOR)
DEFINE_KERNEL(k_cmpu, L_CMPU)
DEFINE_KERNEL(k_lshladd64, L_LSHLADD64)
DEFINE_KERNEL(k_sel, L_SEL)
DEFINE_KERNEL(k_sels, L_SELS)
DEFINE_KERNEL(k_mullo, L_MULLO)
DEFINE_KERNEL(k_mixaddu, L_MIXADDU)
DEFINE_KERNEL(k_mixcnd, L_MIXCND)
DEFINE_KERNEL(k_mixmov64, L_MIXMOV64)
DEFINE_KERNEL(k_mixmullo, L_MIXMULLO)
DEFINE_KERNEL(k_mixcmp, L_MIXCMP)

typedef void (*kern_t)(double *, int, unsigned long long *, int);

int main()
{
    struct { const char *name; kern_t k; double per_line; } classes[] = {
        {"v_fma_f64", k_fma, 1.0}, {"v_mul_f64", k_mul, 1.0}, {"v_add_f64", k_add, 1.0}, {"v_rcp_f64", k_rcp, 1.0}, {"v_rsq_f64", k_rsq, 1.0},
        {"v_max_f64", k_max, 1.0}, {"v_cmp_lt_f64", k_cmp, 1.0}, {"v_div_fixup_f64", k_fixup, 1.0}, {"v_ldexp_f64", k_ldexp, 1.0},
        {"v_rndne_f64", k_rndne, 1.0}, {"v_cmp_class_f64", k_class, 1.0}, {"v_mov_b64", k_mov64, 1.0}, {"v_mov_b32", k_mov32, 1.0},
        {"v_cndmask_b32", k_cnd, 1.0}, {"v_add_u32", k_addu, 1.0}, {"v_xor_b32", k_xor, 1.0}, {"v_cmp_eq_u32", k_cmpu, 1.0},
        {"v_lshl_add_u64", k_lshladd64, 1.0},
        {"select of a double: v_cmp_lt_f64 vcc + 2 x v_cndmask_b32 (per instruction, 3 per select)", k_sel, 3.0},
        {"the same with the mask in an SGPR pair (VOP3)", k_sels, 3.0},
        {"v_mul_lo_u32", k_mullo, 1.0},
        {"MIX v_fma_f64 + v_add_u32 (per instruction, 2 per line)", k_mixaddu, 2.0},
        {"MIX v_fma_f64 + v_cndmask_b32 (per instruction)", k_mixcnd, 2.0},
        {"MIX v_fma_f64 + v_mov_b64 (per instruction)", k_mixmov64, 2.0},
        {"MIX v_fma_f64 + v_mul_lo_u32 (per instruction)", k_mixmullo, 2.0},
        {"MIX v_fma_f64 + v_cmp_lt_f64 (per instruction)", k_mixcmp, 2.0}};
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int simds = prop.multiProcessorCount * 4;
    double *d_out;
    unsigned long long *d_clk, h_clk[2];
    (void)hipMalloc(&d_out, sizeof(double) * 64 * simds * 4);
    (void)hipMalloc(&d_clk, 16);
    const int iters = 2000;                                 // x 256 instructions per wave
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("{\"simds\": %d, \"instructions_per_wave\": %d, \"note\": \"cycles per wave64 instruction per SIMD = kernel time x in-kernel clock x SIMDs / (waves x instructions per wave); 8 independent streams\", \"classes\": {", simds, iters * 256);
    for (size_t c = 0; c < sizeof(classes) / sizeof(classes[0]); ++c) {
        printf("%s\"%s\": {", c ? ", " : "", classes[c].name);
        for (int wps = 1; wps <= 3; ++wps) {
            const int waves = simds * wps;
            float best = 1e30f;
            double ghz = 0.0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(classes[c].k, dim3(waves), dim3(64), 0, 0, d_out, iters, d_clk, 64);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                (void)hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost);
                if (ms < best) { best = ms; ghz = (double)h_clk[0] / (double)h_clk[1] * 0.1; }
            }
            const double per_line = classes[c].per_line;
            const double cycles = best * 1e-3 * ghz * 1e9 * simds / ((double)waves * iters * 256.0 * per_line);
            printf("%s\"waves_per_simd_%d\": {\"cycles_per_inst\": %.2f, \"ms\": %.3f, \"clock_GHz\": %.3f}", wps > 1 ? ", " : "", wps, cycles, best, ghz);
        }
        printf("}");
    }
    printf("}, \"sparse_exec_3_waves_per_simd\": {");
    {   // does an instruction cost less when only some lanes are active?  (lanes [0, active) of every wave)
        struct { const char *name; kern_t k; double per_line; } sp[] = {{"v_fma_f64", k_fma, 1.0}, {"v_mul_f64", k_mul, 1.0}, {"v_add_u32", k_addu, 1.0},
                                                                      {"v_mov_b64", k_mov64, 1.0}, {"v_rcp_f64", k_rcp, 1.0}, {"v_cmp_lt_f64", k_cmp, 1.0}};
        const int act[] = {1, 8, 16, 17, 32, 33, 48, 64};
        for (size_t c = 0; c < sizeof(sp) / sizeof(sp[0]); ++c) {
            printf("%s\"%s\": {", c ? ", " : "", sp[c].name);
            for (size_t ai = 0; ai < sizeof(act) / sizeof(act[0]); ++ai) {
                const int waves = simds * 3;
                float best = 1e30f; double ghz = 0.0;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipEventRecord(e0);
                    hipLaunchKernelGGL(sp[c].k, dim3(waves), dim3(64), 0, 0, d_out, iters, d_clk, act[ai]);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    (void)hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost);
                    if (ms < best) { best = ms; ghz = (double)h_clk[0] / (double)h_clk[1] * 0.1; }
                }
                const double cycles = best * 1e-3 * ghz * 1e9 * simds / ((double)waves * iters * 256.0 * sp[c].per_line);
                printf("%s\"active_%d\": %.2f", ai ? ", " : "", act[ai], cycles);
            }
            printf("}");
        }
    }
    printf("}}\n");
    return 0;
}
