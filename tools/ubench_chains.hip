// Micro-benchmark: v_mfma_f32_32x32x16_bf16 issue rate vs number of INDEPENDENT accumulator chains per wave
// and waves per SIMD.  ns per MFMA per SIMD (lower bound 32 cycles ~ 14.5 ns at 2.2 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITER 4000
template <int CH>
__global__ void k(float* out, const uint4* in) {
    const uint4 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 16 / CH; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[c], 0, 0, 0);
    }
    float r = 0.f;
    for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) r += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int CH> float run(int wps, float* out, const uint4* in) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<CH><<<256 * wps, 256>>>(out, in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<CH><<<256 * wps, 256>>>(out, in); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / ITER / 16.0f / wps;  // ns per MFMA per SIMD
}
int main() {
    float* out; uint4* in;
    (void)hipMalloc(&out, 256 * 8 * 256 * 4 * 2); (void)hipMalloc(&in, 128 * sizeof(uint4));
    std::vector<uint32_t> h(512, 0x3c003c00u); (void)hipMemcpy(in, h.data(), 2048, hipMemcpyHostToDevice);
    printf("ns per MFMA per SIMD (32 cycles = %.1f ns at 2.2 GHz)\n", 32 / 2.2);
    for (int w = 1; w <= 4; w *= 2)
        printf("waves/SIMD %d: 1 chain %.1f | 2 chains %.1f | 4 chains %.1f\n", w, run<1>(w, out, in), run<2>(w, out, in), run<4>(w, out, in));
    return 0;
}
