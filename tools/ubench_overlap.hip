// Micro-benchmark: how do MFMA chains and VALU (fma/exp/add) streams share a gfx950 SIMD?
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_overlap.hip -o /tmp/ubench && /tmp/ubench
// One workgroup of (64 * waves_per_simd * 4) threads per CU-slot, grid = 256 * blocks_per_cu.
// Each kernel runs ITER iterations of a fixed per-wave work unit:
//   unit = 16 MFMAs (two dependent chains of 8, v_mfma_f32_32x32x16_bf16) and/or 32 x (fma, exp2, add)
// Reported: ns per iteration per wave-slot, i.e. elapsed / ITER, for different waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ITER 2000

template <int MODE>  // 0 mfma only, 1 valu only, 2 sequential (mfma then valu), 3 interleaved
__global__ void k(float* out, const uint4* in) {
    const uint4 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x16 acc0, acc1;
    float v[32];
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int i = 0; i < 32; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
    float s0 = 0.f, s1 = 0.f;
    const float c = 0.127f, off = -0.5f;
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc1, 0, 0, 0);
            }
        }
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 32; j += 2) {
                s0 += __builtin_amdgcn_exp2f(fmaf(v[j], c, off));
                s1 += __builtin_amdgcn_exp2f(fmaf(v[j + 1], c, off));
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
                s0 += __builtin_amdgcn_exp2f(fmaf(v[4 * j], c, off));
                s1 += __builtin_amdgcn_exp2f(fmaf(v[4 * j + 1], c, off));
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc1, 0, 0, 0);
                s0 += __builtin_amdgcn_exp2f(fmaf(v[4 * j + 2], c, off));
                s1 += __builtin_amdgcn_exp2f(fmaf(v[4 * j + 3], c, off));
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
            }
        }
        // ALL VALU inputs change every iteration (with only a few changing, the compiler hoists the invariant
        // fma/exp work out of the loop and the benchmark under-counts VALU time -- that happened in round 1)
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] += 1e-6f;
    }
    float r = s0 + s1;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
float run(int waves_per_simd, float* out, const uint4* in) {
    const int threads = 256;  // 4 waves: one per SIMD
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, in);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / ITER;  // ns per iteration
}

int main() {
    float* out; uint4* in;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 2);
    hipMalloc(&in, 128 * sizeof(uint4));
    std::vector<uint32_t> h(128 * 4, 0x3c003c00u);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"mfma only (16 MFMA 32x32x16, 2 chains)", "valu only (32 x fma+exp2+add)", "sequential mfma;valu", "interleaved"};
    printf("ns per iteration (all waves run concurrently; unit = 16 MFMAs and/or 96 VALU)\n");
    for (int w = 1; w <= 4; w *= 2) {
        float t0 = run<0>(w, out, in), t1 = run<1>(w, out, in), t2 = run<2>(w, out, in), t3 = run<3>(w, out, in);
        printf("waves/SIMD %d: %s %.0f | %s %.0f | %s %.0f | %s %.0f   (sum %.0f, max %.0f)\n", w, names[0], t0, names[1], t1, names[2],
               t2, names[3], t3, t0 + t1, t0 > t1 ? t0 : t1);
    }
    return 0;
}
