// valu_issue.hip -- how many vector instructions per cycle one SIMD of an MI355X issues, by instruction
// kind, by the number of resident waves per SIMD and by the number of independent dependency chains
// inside a wave.  Every CU runs the same loop: 64 instructions of one kind per iteration over 1..8
// accumulator chains (inline asm, so the compiler neither removes nor reorders them).
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue > valu_issue.csv
//
// Output columns: op, waves_per_simd, shader_MHz, cyc_per_instr_wave (cycles a wave spends per
// instruction, from s_memtime inside the wave), instr_per_cyc_simd (all waves of a SIMD together, from
// the kernel's wall time and the measured clock), lane_results_per_cyc_simd (x2 for packed ops).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long cyc, wall; };
typedef float f2 __attribute__((ext_vector_type(2)));

#define R8(x) x x x x x x x x

// BODY8: eight instructions over accumulators %0..%7, inputs %8 %9 (vector) and %10 (scalar).
#define KERNEL(name, BODY8, ...)                                                                            \
    __global__ __launch_bounds__(256) void name(float *out, int iters, Stamp *stamps) {                     \
        extern __shared__ float lds[];                                                                     \
        if (iters < 0) lds[threadIdx.x] = 1.0f;                                                            \
        float t = (float)threadIdx.x * 1e-3f;                                                              \
        float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;  \
        float b = 1.0001f + t * 1e-7f, c = 0.25f + t * 1e-7f;                                               \
        float s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0002f + (float)iters * 1e-9f))); \
        unsigned long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime();                          \
        for (int i = 0; i < iters; i++) {                                                                  \
            asm volatile(R8(BODY8)                                                                         \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)  \
                         : "v"(b), "v"(c), "s"(s)                                                          \
                         : __VA_ARGS__);                                                                   \
        }                                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                 \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            Stamp st = {t1 - t0, w1 - w0};                                                                 \
            stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;                                     \
        }                                                                                                  \
    }
// the same over eight register PAIRS (packed FP32)
#define KERNEL_PK(name, BODY8)                                                                              \
    __global__ __launch_bounds__(256) void name(float *out, int iters, Stamp *stamps) {                     \
        extern __shared__ float lds[];                                                                     \
        if (iters < 0) lds[threadIdx.x] = 1.0f;                                                            \
        float t = (float)threadIdx.x * 1e-3f;                                                              \
        f2 a0 = {t, t + 8}, a1 = {t + 1, t + 9}, a2 = {t + 2, t + 10}, a3 = {t + 3, t + 11};               \
        f2 a4 = {t + 4, t + 12}, a5 = {t + 5, t + 13}, a6 = {t + 6, t + 14}, a7 = {t + 7, t + 15};         \
        f2 b = {1.0001f + t * 1e-7f, 1.0002f}, c = {0.25f + t * 1e-7f, 0.125f};                            \
        unsigned long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime();                          \
        for (int i = 0; i < iters; i++) {                                                                  \
            asm volatile(R8(BODY8)                                                                         \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)  \
                         : "v"(b), "v"(c)                                                                  \
                         : "memory");                                                                      \
        }                                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();                          \
        f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;                                            \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            Stamp st = {t1 - t0, w1 - w0};                                                                 \
            stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;                                     \
        }                                                                                                  \
    }

#define I3(op, x, y) op " %0, %0, " x ", " y "\n " op " %1, %1, " x ", " y "\n " op " %2, %2, " x ", " y "\n " op " %3, %3, " x ", " y "\n " \
                     op " %4, %4, " x ", " y "\n " op " %5, %5, " x ", " y "\n " op " %6, %6, " x ", " y "\n " op " %7, %7, " x ", " y "\n "
#define I2(op, x) op " %0, %0, " x "\n " op " %1, %1, " x "\n " op " %2, %2, " x "\n " op " %3, %3, " x "\n " \
                  op " %4, %4, " x "\n " op " %5, %5, " x "\n " op " %6, %6, " x "\n " op " %7, %7, " x "\n "
#define I1(op) op " %0, %0\n " op " %1, %1\n " op " %2, %2\n " op " %3, %3\n " op " %4, %4\n " op " %5, %5\n " op " %6, %6\n " op " %7, %7\n "
// n chains only: the same register every n-th instruction
#define I3_4(op, x, y) op " %0, %0, " x ", " y "\n " op " %1, %1, " x ", " y "\n " op " %2, %2, " x ", " y "\n " op " %3, %3, " x ", " y "\n " \
                       op " %0, %0, " x ", " y "\n " op " %1, %1, " x ", " y "\n " op " %2, %2, " x ", " y "\n " op " %3, %3, " x ", " y "\n "
#define I3_2(op, x, y) op " %0, %0, " x ", " y "\n " op " %1, %1, " x ", " y "\n " op " %0, %0, " x ", " y "\n " op " %1, %1, " x ", " y "\n " \
                       op " %0, %0, " x ", " y "\n " op " %1, %1, " x ", " y "\n " op " %0, %0, " x ", " y "\n " op " %1, %1, " x ", " y "\n "
#define I3_1(op, x, y) op " %0, %0, " x ", " y "\n " op " %0, %0, " x ", " y "\n " op " %0, %0, " x ", " y "\n " op " %0, %0, " x ", " y "\n " \
                       op " %0, %0, " x ", " y "\n " op " %0, %0, " x ", " y "\n " op " %0, %0, " x ", " y "\n " op " %0, %0, " x ", " y "\n "

KERNEL(k_fma_8chains, I3("v_fma_f32", "%8", "%9"), "memory")
KERNEL(k_fma_4chains, I3_4("v_fma_f32", "%8", "%9"), "memory")
KERNEL(k_fma_2chains, I3_2("v_fma_f32", "%8", "%9"), "memory")
KERNEL(k_fma_1chain, I3_1("v_fma_f32", "%8", "%9"), "memory")
KERNEL(k_fma_sgpr_operand, I3("v_fma_f32", "%10", "%9"), "memory")
KERNEL(k_mul, I2("v_mul_f32", "%8"), "memory")
KERNEL(k_add, I2("v_add_f32", "%9"), "memory")
KERNEL(k_max, I2("v_max_f32", "%9"), "memory")
KERNEL(k_max3, I3("v_max3_f32", "%8", "%9"), "memory")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %8\n v_mov_b32 %3, %9\n v_mov_b32 %4, %8\n v_mov_b32 %5, %9\n v_mov_b32 %6, %8\n v_mov_b32 %7, %9\n", "memory")
KERNEL(k_cndmask, I3("v_cndmask_b32", "%8", "vcc"), "memory")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
              "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n", "memory", "vcc")
KERNEL(k_cmp_then_cndmask, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                           "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n", "memory", "vcc")
KERNEL(k_rcp, I1("v_rcp_f32"), "memory")
KERNEL(k_sqrt, I1("v_sqrt_f32"), "memory")
KERNEL(k_rsq, I1("v_rsq_f32"), "memory")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n"
                   "v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 5\n v_readlane_b32 s22, %6, 7\n v_readlane_b32 s23, %7, 9\n",
       "memory", "s20", "s21", "s22", "s23")
// exact division as the compiler emits it: scale, scale, rcp, 4 fma, mul, fma, fmas, fixup -- one chain of 11 dependent instructions
KERNEL(k_div_sequence, "v_div_scale_f32 %1, vcc, %8, %8, %0\n v_div_scale_f32 %2, vcc, %0, %8, %0\n v_rcp_f32 %3, %1\n v_fma_f32 %4, -%1, %3, 1.0\n"
                       "v_fma_f32 %3, %4, %3, %3\n v_mul_f32 %4, %2, %3\n v_fma_f32 %5, -%1, %4, %2\n v_fma_f32 %4, %5, %3, %4\n",
       "memory", "vcc")
// vector and scalar instructions of one wave interleaved: does a scalar instruction cost a vector issue slot?
KERNEL(k_fma_with_salu, "v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s21, s21, 1\n"
                        "v_fma_f32 %2, %2, %8, %9\n s_add_u32 s22, s22, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s23, s23, 1\n",
       "memory", "s20", "s21", "s22", "s23", "scc")
// select with a mask in an SGPR pair that a compare wrote once before the loop / inside the loop
KERNEL(k_cndmask_sgpr_mask, "v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                            "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n",
       "memory", "s20", "s21")
KERNEL(k_cmp_sgpr_then_cndmask, "v_cmp_lt_f32 s[20:21], %0, %8\n v_cndmask_b32 %0, %0, %9, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cndmask_b32 %1, %1, %9, s[22:23]\n"
                                "v_cmp_lt_f32 s[20:21], %2, %8\n v_cndmask_b32 %2, %2, %9, s[20:21]\n v_cmp_lt_f32 s[22:23], %3, %8\n v_cndmask_b32 %3, %3, %9, s[22:23]\n",
       "memory", "s20", "s21", "s22", "s23")
// a mix close to the renderer's inner loops: mul, fma, add, sub, max, min, compare, select
KERNEL(k_mix, "v_mul_f32 %0, %0, %8\n v_fma_f32 %1, %1, %8, %9\n v_add_f32 %2, %2, %9\n v_max_f32 %3, %3, %9\n"
              "v_sub_f32 %4, %4, %9\n v_min_f32 %5, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n", "memory", "vcc")

// scalar accumulators %0..%7 (SGPRs), vector accumulators %8..%11, vector inputs %12 %13
#define KERNEL_S(name, BODY8, ...)                                                                          \
    __global__ __launch_bounds__(256) void name(float *out, int iters, Stamp *stamps) {                     \
        extern __shared__ float lds[];                                                                     \
        if (iters < 0) lds[threadIdx.x] = 1.0f;                                                            \
        int u = __builtin_amdgcn_readfirstlane(iters);                                                     \
        int s0 = u, s1 = u + 1, s2 = u + 2, s3 = u + 3, s4 = u + 4, s5 = u + 5, s6 = u + 6, s7 = u + 7;    \
        float t = (float)threadIdx.x * 1e-3f;                                                              \
        float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3;                                                   \
        float b = 1.0001f + t * 1e-7f, c = 0.25f + t * 1e-7f;                                               \
        unsigned long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime();                          \
        for (int i = 0; i < iters; i++) {                                                                  \
            asm volatile(R8(BODY8)                                                                         \
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7), \
                           "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                          \
                         : "v"(b), "v"(c)                                                                  \
                         : __VA_ARGS__);                                                                   \
        }                                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7); \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            Stamp st = {t1 - t0, w1 - w0};                                                                 \
            stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;                                     \
        }                                                                                                  \
    }
KERNEL_S(k_salu_add, "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                     "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n", "memory", "scc")
KERNEL_S(k_salu_and64, "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_or_b64 s[28:29], s[28:29], s[22:23]\n"
                       "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_or_b64 s[28:29], s[28:29], s[22:23]\n",
         "memory", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29")
KERNEL_S(k_salu_mov, "s_mov_b32 %0, %1\n s_mov_b32 %2, %3\n s_mov_b32 %4, %5\n s_mov_b32 %6, %7\n"
                     "s_mov_b32 %1, %0\n s_mov_b32 %3, %2\n s_mov_b32 %5, %4\n s_mov_b32 %7, %6\n", "memory")
// branches: not taken (the compare is never true) and taken (to the next instruction)
KERNEL_S(k_branch_not_taken, "s_cmp_eq_u32 %0, -7\n s_cbranch_scc1 9f\n s_cmp_eq_u32 %1, -7\n s_cbranch_scc1 9f\n"
                             "s_cmp_eq_u32 %2, -7\n s_cbranch_scc1 9f\n s_cmp_eq_u32 %3, -7\n s_cbranch_scc1 9f\n 9:\n", "memory", "scc")
KERNEL_S(k_branch_taken, "s_branch 1f\n 1: s_branch 2f\n 2: s_branch 3f\n 3: s_branch 4f\n 4: s_branch 5f\n 5: s_branch 6f\n 6: s_branch 7f\n 7: s_branch 8f\n 8:\n", "memory")
// vector : scalar = 1:1, 2:1, 3:1 (eight instructions per body; the vector ones are counted)
KERNEL_S(k_v1s1, "v_fma_f32 %8, %8, %12, %13\n s_add_u32 %0, %0, 1\n v_fma_f32 %9, %9, %12, %13\n s_add_u32 %1, %1, 1\n"
                 "v_fma_f32 %10, %10, %12, %13\n s_add_u32 %2, %2, 1\n v_fma_f32 %11, %11, %12, %13\n s_add_u32 %3, %3, 1\n", "memory", "scc")
KERNEL_S(k_v3s1, "v_fma_f32 %8, %8, %12, %13\n v_fma_f32 %9, %9, %12, %13\n v_fma_f32 %10, %10, %12, %13\n s_add_u32 %0, %0, 1\n"
                 "v_fma_f32 %11, %11, %12, %13\n v_fma_f32 %8, %8, %12, %13\n v_fma_f32 %9, %9, %12, %13\n s_add_u32 %1, %1, 1\n", "memory", "scc")
KERNEL_S(k_v6s2_grouped, "v_fma_f32 %8, %8, %12, %13\n v_fma_f32 %9, %9, %12, %13\n v_fma_f32 %10, %10, %12, %13\n v_fma_f32 %11, %11, %12, %13\n"
                         "v_fma_f32 %8, %8, %12, %13\n v_fma_f32 %9, %9, %12, %13\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n", "memory", "scc")
// what a packet-traversal step does besides arithmetic: compare -> mask in an SGPR pair -> scalar logic on it
KERNEL_S(k_cmp_ballot_salu, "v_cmp_lt_f32 s[20:21], %8, %12\n s_and_b64 s[22:23], s[20:21], exec\n v_cmp_lt_f32 s[24:25], %9, %12\n s_or_b64 s[22:23], s[22:23], s[24:25]\n"
                            "v_cmp_lt_f32 s[20:21], %10, %12\n s_and_b64 s[22:23], s[20:21], exec\n v_cmp_lt_f32 s[24:25], %11, %12\n s_or_b64 s[22:23], s[22:23], s[24:25]\n",
         "memory", "scc", "s20", "s21", "s22", "s23", "s24", "s25")
KERNEL_S(k_readfirstlane, "v_readfirstlane_b32 %0, %8\n v_readfirstlane_b32 %1, %9\n v_readfirstlane_b32 %2, %10\n v_readfirstlane_b32 %3, %11\n"
                          "v_readfirstlane_b32 %4, %8\n v_readfirstlane_b32 %5, %9\n v_readfirstlane_b32 %6, %10\n v_readfirstlane_b32 %7, %11\n", "memory")
KERNEL_PK(k_pk_fma, I3("v_pk_fma_f32", "%8", "%9"))
KERNEL_PK(k_pk_mul, I2("v_pk_mul_f32", "%8"))
KERNEL_PK(k_pk_add, I2("v_pk_add_f32", "%9"))

typedef void (*kern_t)(float *, int, Stamp *);
struct Op { const char *name; kern_t k; int instr_per_body; int results; };

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    fprintf(stderr, "%s, %d CUs, %d MHz max\n", prop.name, cus, prop.clockRate / 1000);
    float *out;
    Stamp *stamps;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 6 * 256 * sizeof(float)));
    CHECK(hipMalloc(&stamps, (size_t)cus * 8 * 6 * 4 * sizeof(Stamp)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const Op ops[] = {
        {"v_fma_f32 x8 chains", k_fma_8chains, 64, 1}, {"v_fma_f32 x4 chains", k_fma_4chains, 64, 1},
        {"v_fma_f32 x2 chains", k_fma_2chains, 64, 1}, {"v_fma_f32 x1 chain", k_fma_1chain, 64, 1},
        {"v_fma_f32 sgpr operand", k_fma_sgpr_operand, 64, 1}, {"v_mul_f32", k_mul, 64, 1}, {"v_add_f32", k_add, 64, 1},
        {"v_max_f32", k_max, 64, 1}, {"v_max3_f32", k_max3, 64, 1}, {"v_mov_b32", k_mov, 64, 1},
        {"v_cndmask_b32", k_cndmask, 64, 1}, {"v_cmp_lt_f32", k_cmp, 64, 1}, {"v_cmp+v_cndmask pairs", k_cmp_then_cndmask, 64, 1},
        {"v_rcp_f32", k_rcp, 64, 1}, {"v_sqrt_f32", k_sqrt, 64, 1}, {"v_rsq_f32", k_rsq, 64, 1},
        {"v_readlane_b32", k_readlane, 64, 1}, {"div sequence (8 dependent)", k_div_sequence, 64, 1},
        {"v_fma_f32 + s_add_u32 interleaved (vector instr counted)", k_fma_with_salu, 32, 1},
        {"v_cndmask_b32 sgpr-pair mask", k_cndmask_sgpr_mask, 64, 1}, {"v_cmp(sgpr pair)+v_cndmask pairs", k_cmp_sgpr_then_cndmask, 64, 1},
        {"mix mul/fma/add/max/sub/min/cmp/cndmask", k_mix, 64, 1},
        {"s_add_u32 (scalar only)", k_salu_add, 64, 1}, {"s_and_b64/s_or_b64 (scalar only)", k_salu_and64, 64, 1}, {"s_mov_b32 (scalar only)", k_salu_mov, 64, 1},
        {"s_cmp+s_cbranch not taken (both counted)", k_branch_not_taken, 64, 1}, {"s_branch taken", k_branch_taken, 64, 1},
        {"v_fma : s_add 1:1 (vector counted)", k_v1s1, 32, 1}, {"v_fma : s_add 3:1 (vector counted)", k_v3s1, 48, 1},
        {"v_fma x6 then s_add x2 (vector counted)", k_v6s2_grouped, 48, 1},
        {"v_cmp->sgpr pair + s_and/s_or_b64 1:1 (vector counted)", k_cmp_ballot_salu, 32, 1},
        {"v_readfirstlane_b32", k_readfirstlane, 64, 1},
        {"v_pk_fma_f32", k_pk_fma, 64, 2}, {"v_pk_mul_f32", k_pk_mul, 64, 2}, {"v_pk_add_f32", k_pk_add, 64, 2},
    };
    // Blocks of 256 threads (one wave per SIMD); W of them fit a CU because each asks for 1/W of its 160 KiB
    // of LDS, and the grid holds 6 rounds of them, so that every SIMD has W resident waves throughout.
    const int wps_list[] = {1, 2, 4, 6, 8};
    printf("op,waves_per_simd,shader_MHz,cyc_per_instr_wave,instr_per_cyc_simd,lane_results_per_cyc_simd\n");
    for (const Op &op : ops) {
        CHECK(hipFuncSetAttribute((const void *)op.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int wps : wps_list) {
            int blocks = cus * wps * 6, threads = 256;
            size_t lds = (size_t)(160 * 1024 / wps) / 1024 * 1024 - (wps > 1 ? 1024 : 0);
            int waves = blocks * threads / 64;
            for (int rep = 0; rep < 2; rep++) {  // the first run warms up clocks and code
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(op.k, dim3(blocks), dim3(threads), lds, 0, out, iters, stamps);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
            }
            CHECK(hipGetLastError());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Stamp> st(waves);
            CHECK(hipMemcpy(st.data(), stamps, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
            double cyc = 0, wall = 0;
            for (const Stamp &s : st) { cyc += (double)s.cyc; wall += (double)s.wall; }
            cyc /= waves; wall /= waves;                 // per wave: shader cycles, ticks of the 100 MHz clock
            double mhz = cyc / wall * 100.0;
            double n_instr = (double)iters * op.instr_per_body;  // counted vector instructions per wave
            double per_wave = cyc / n_instr;
            double simd_rate = n_instr * waves / (cus * 4.0) / (ms * 1e-3 * mhz * 1e6);
            printf("%s,%d,%.0f,%.3f,%.3f,%.3f\n", op.name, wps, mhz, per_wave, simd_rate, simd_rate * op.results);
            fflush(stdout);
        }
    }
    return 0;
}
