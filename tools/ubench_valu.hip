// Micro-benchmark: issue cost of the VALU instructions the softmax passes are made of, on a gfx950 SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
// Every wave runs ITER iterations of 64 instructions of one kind on 16 independent registers (so the
// dependency distance is 16 instructions); 1, 2 or 4 waves per SIMD.  Reported: SIMD cycles per wave64
// instruction = elapsed * clock / (ITER * 64 * waves_per_simd), clock from hipDeviceProp (current, not boost).
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 4096
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { OP_FMA, OP_ADD, OP_MAX3, OP_EXP, OP_MUL, OP_PKFMA, OP_PKADD, OP_CVTPK, OP_LOG, OP_RCP, OP_FMA_EXP_ADD, OP_DPP, OP_EXPLEG, OP_EXPF16, OP_N };

template <int OP>
__global__ void k(float* out, float seed) {
    float v[16], s[16];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = seed * (float)(threadIdx.x + i + 1); s[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { p[i][0] = v[2 * i]; p[i][1] = v[2 * i + 1]; }
    const float c = 0.999f, d = 1e-7f;
    const f32x2 c2 = {c, c}, d2 = {d, d};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
                if (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(d));
                if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
                if (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
                if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if (OP == OP_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(v[i]));
                if (OP == OP_EXPLEG) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(v[i]));
                if (OP == OP_EXPF16) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
                if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                if (OP == OP_PKFMA && i < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(d2));
                if (OP == OP_PKADD && i < 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(d2));
                if (OP == OP_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
                if (OP == OP_DPP) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
                if (OP == OP_FMA_EXP_ADD) {  // the softmax triple: 3 instructions, counted as 3
                    float t;
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v[i]), "v"(c), "v"(d));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(t));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(t));
                }
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += v[i] + s[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>  // 0: 32x32x16 bf16 (16 acc regs), 1: 16x16x32 bf16 (4 acc regs); two independent chains per wave
__global__ void km(float* out, const uint4* in) {
    const uint4 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x16 A0 = {}, A1 = {};
    f32x4 B0 = {}, B1 = {};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (SHAPE == 0) {
                A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), A0, 0, 0, 0);
                A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), A1, 0, 0, 0);
            } else {
                B0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), B0, 0, 0, 0);
                B1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), B1, 0, 0, 0);
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += A0[i] + A1[i];
    for (int i = 0; i < 4; ++i) r += B0[i] + B1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int SHAPE>
float runm(int wps, float* out, const uint4* in, double ghz) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int threads = 256 * wps;
    km<SHAPE><<<256, threads>>>(out, in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); km<SHAPE><<<256, threads>>>(out, in); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return (float)(ms * 1e6 * ghz / ((double)ITER * 16 * wps));
}

template <int OP>
float run(int wps, float* out, double ghz) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int threads = 256 * wps;  // wps waves on each of the 4 SIMDs, one workgroup per CU
    k<OP><<<256, threads>>>(out, 1e-3f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<OP><<<256, threads>>>(out, 1e-3f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    int per_iter = 64;
    if (OP == OP_PKFMA || OP == OP_PKADD) per_iter = 32;
    if (OP == OP_FMA_EXP_ADD) per_iter = 192;
    return (float)(ms * 1e6 * ghz / ((double)ITER * per_iter * wps));
}

int main() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const double ghz = pr.clockRate * 1e-6;
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    printf("clock %.3f GHz; SIMD cycles per wave64 instruction at 1 / 2 / 4 waves per SIMD\n", ghz);
    const char* names[OP_N] = {"v_fma_f32", "v_add_f32", "v_max3_f32", "v_exp_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_add_f32",
                               "v_cvt_pk_bf16_f32", "v_log_f32", "v_rcp_f32", "fma+exp+add (per instr)", "v_add_f32_dpp", "v_exp_legacy_f32", "v_exp_f16"};
#define ROW(OP) printf("%-26s %6.2f %6.2f %6.2f\n", names[OP], run<OP>(1, out, ghz), run<OP>(2, out, ghz), run<OP>(4, out, ghz));
    ROW(OP_FMA) ROW(OP_ADD) ROW(OP_MUL) ROW(OP_MAX3) ROW(OP_EXP) ROW(OP_LOG) ROW(OP_RCP) ROW(OP_PKFMA) ROW(OP_PKADD) ROW(OP_CVTPK)
    ROW(OP_FMA_EXP_ADD) ROW(OP_DPP) ROW(OP_EXPLEG) ROW(OP_EXPF16)
    uint4* in; (void)hipMalloc(&in, 128 * sizeof(uint4)); (void)hipMemset(in, 0x3c, 128 * sizeof(uint4));
    printf("%-26s %6.2f %6.2f %6.2f\n", "v_mfma_f32_32x32x16_bf16", runm<0>(1, out, in, ghz), runm<0>(2, out, in, ghz), runm<0>(4, out, in, ghz));
    printf("%-26s %6.2f %6.2f %6.2f\n", "v_mfma_f32_16x16x32_bf16", runm<1>(1, out, in, ghz), runm<1>(2, out, in, ghz), runm<1>(4, out, in, ghz));
    return 0;
}
