// Issue cost of the VALU instruction classes march_kernel is made of (profiles/r05_instr_mix.json), measured: for each opcode a kernel of
// 8 independent dependency chains per lane (no memory, 4 waves per SIMD resident on every CU) is timed against the same loop of v_fma_f64,
// whose rate the data sheet fixes (78.6 TFLOP/s = one wave64 fp64 FMA per 4 cycles per SIMD).  cycles(op) = 4 x t(op) / t(v_fma_f64).
// hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 4096;

#define CHAIN8(ASM64)                                                                                          \
    for (int i = 0; i < ITER; ++i) {                                                                           \
        asm volatile(ASM64(0) ASM64(1) ASM64(2) ASM64(3) ASM64(4) ASM64(5) ASM64(6) ASM64(7)                    \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
    }
#define OP_FMA64(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
#define OP_ADD64(k) "v_add_f64 %" #k ", %" #k ", %8\n"
#define OP_MUL64(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
#define OP_MAX64(k) "v_max_f64 %" #k ", %" #k ", %8\n"
#define OP_FRACT64(k) "v_fract_f64 %" #k ", %" #k "\n"

__device__ long long* g_cyc;
#define T0 const long long c0_ = clock64();
#define T1 if ((threadIdx.x & 63) == 0) g_cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = clock64() - c0_;
template <int OP> __global__ __launch_bounds__(256) void k64(double* out, double b, double c) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    T0
    if (OP == 0) { CHAIN8(OP_FMA64) } else if (OP == 1) { CHAIN8(OP_ADD64) } else if (OP == 2) { CHAIN8(OP_MUL64) } else if (OP == 3) { CHAIN8(OP_MAX64) } else { CHAIN8(OP_FRACT64) }
    T1
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// 32-bit / mixed classes: operands as separate 32- and 64-bit registers
#define CHAIN8M(STMT) for (int i = 0; i < ITER; ++i) { STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) }
template <int OP> __global__ __launch_bounds__(256) void kmix(double* out, float fb, int ib) {
    float f[8]; double d[8]; int n[8];
    for (int k = 0; k < 8; ++k) { f[k] = threadIdx.x + k; d[k] = threadIdx.x + 0.5 * k; n[k] = threadIdx.x + k; }
#define S_CVT_F64_F32(k) asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0\n" : "+v"(d[k]), "+v"(f[k]));          /* a round trip: two conversions */
#define S_CVT_I32_F64(k) asm volatile("v_cvt_i32_f64 %1, %0\n v_cvt_f64_i32 %0, %1\n" : "+v"(d[k]), "+v"(n[k]));
#define S_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %1\n" : "+v"(f[k]) : "v"(fb));
#define S_ADDU32(k) asm volatile("v_add_u32 %0, %0, %1\n" : "+v"(n[k]) : "v"(ib));
#define S_MOV(k) asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n" : "+v"(n[k]), "+v"(n[(k + 1) & 7]));
#define S_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n" : "+v"(n[k]) : "v"(ib) : "vcc");
#define S_CMP64(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n" : : "v"(d[k]), "v"(d[(k + 1) & 7]) : "vcc");
#define S_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1\n" : "+v"(n[k]) : "v"(ib));
#define S_MAD24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n" : "+v"(n[k]) : "v"(ib));
#define S_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n" : "+v"(d[k]));
    T0
    if (OP == 0) { CHAIN8M(S_CVT_F64_F32) } else if (OP == 1) { CHAIN8M(S_CVT_I32_F64) } else if (OP == 2) { CHAIN8M(S_FMA32) } else if (OP == 3) { CHAIN8M(S_ADDU32) }
    else if (OP == 4) { CHAIN8M(S_MOV) } else if (OP == 5) { CHAIN8M(S_CNDMASK) } else if (OP == 6) { CHAIN8M(S_CMP64) } else if (OP == 7) { CHAIN8M(S_MULLO) }
    else if (OP == 8) { CHAIN8M(S_MAD24) } else { CHAIN8M(S_PKFMA) }
    T1
    double s = 0; for (int k = 0; k < 8; ++k) s += d[k] + f[k] + n[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> static float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    const int G = 256 * 4;          // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    double* out; CK(hipMalloc(&out, (size_t)G * 256 * 8));
    long long* dcyc; CK(hipMalloc(&dcyc, (size_t)G * 4 * 8)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), &dcyc, sizeof(dcyc)));
    static long long hc[256 * 4 * 4]; double cyc[16];
    const double per = 8.0;        // instructions per chain pass
    float t[16]; const char* name[16]; int n = 0; double mult[16];
#define RUN(NAME, M, ...) name[n] = NAME; mult[n] = M; t[n] = timeit([&] { __VA_ARGS__; }); CK(hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost)); \
    { double s_ = 0; for (int i_ = 0; i_ < G * 4; ++i_) s_ += hc[i_]; cyc[n] = s_ / (G * 4) / ((double)ITER * 8 * M * 4); } ++n;
    RUN("v_fma_f64", 1, hipLaunchKernelGGL(k64<0>, dim3(G), dim3(256), 0, 0, out, 1.0000001, 1e-9))
    RUN("v_add_f64", 1, hipLaunchKernelGGL(k64<1>, dim3(G), dim3(256), 0, 0, out, 1.0000001, 1e-9))
    RUN("v_mul_f64", 1, hipLaunchKernelGGL(k64<2>, dim3(G), dim3(256), 0, 0, out, 1.0000001, 1e-9))
    RUN("v_max_f64", 1, hipLaunchKernelGGL(k64<3>, dim3(G), dim3(256), 0, 0, out, 1.0000001, 1e-9))
    RUN("v_fract_f64", 1, hipLaunchKernelGGL(k64<4>, dim3(G), dim3(256), 0, 0, out, 1.0000001, 1e-9))
    RUN("v_cvt_f64_f32 + v_cvt_f32_f64", 2, hipLaunchKernelGGL(kmix<0>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    RUN("v_cvt_i32_f64 + v_cvt_f64_i32", 2, hipLaunchKernelGGL(kmix<1>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    RUN("v_fma_f32", 1, hipLaunchKernelGGL(kmix<2>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    RUN("v_add_u32", 1, hipLaunchKernelGGL(kmix<3>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    RUN("v_mov_b32 x2", 2, hipLaunchKernelGGL(kmix<4>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    RUN("v_cndmask_b32", 1, hipLaunchKernelGGL(kmix<5>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    RUN("v_cmp_lt_f64", 1, hipLaunchKernelGGL(kmix<6>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    RUN("v_mul_lo_u32", 1, hipLaunchKernelGGL(kmix<7>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    RUN("v_mad_u32_u24", 1, hipLaunchKernelGGL(kmix<8>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    RUN("v_pk_fma_f32", 1, hipLaunchKernelGGL(kmix<9>, dim3(G), dim3(256), 0, 0, out, 1.0001f, 3))
    const double ref = t[0] / (per * 1);
    printf("{\"method\": \"8 independent chains x %d passes per lane, 4 waves per SIMD on 256 CUs, HIP events; cycles = 4 x time per instruction / time per v_fma_f64\", \"cycles_per_wave64_instruction\": {", ITER);
    for (int i = 0; i < n; ++i) printf("%s\"%s\": %.2f", i ? ", " : "", name[i], 4.0 * (t[i] / (per * mult[i])) / ref);
    printf("}, \"shader_cycles_per_wave64_instruction_per_SIMD (s_memtime, 4 waves per SIMD interleaved)\": {");
    for (int i = 0; i < n; ++i) printf("%s\"%s\": %.2f", i ? ", " : "", name[i], cyc[i]);
    const double inst = (double)G * 4 /*waves per WG*/ * ITER * per;
    printf("}, \"v_fma_f64_ms\": %.4f, \"v_fma_f64_G_wave_instr_per_s\": %.1f}\n", t[0], inst / (t[0] * 1e-3) / 1e9);
    return 0;
}
