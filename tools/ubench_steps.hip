// Micro-benchmark: grow a synthetic loop step by step toward snapkv_p1's inner loop and watch where the
// MFMA / VALU overlap is lost.  One 512-thread workgroup per CU (8 waves, 2 per SIMD), ITER "sub-tiles" each:
//   V0  8 dependent MFMAs (32x32x16 bf16)                                   -> matrix-pipe floor
//   V1  V0 + softmax-like VALU on the PREVIOUS sub-tile's accumulators (software pipelined, 2 acc sets)
//   V2  V1 with the MFMA A operands read from LDS (ds_read_b128, prefetched one sub-tile ahead)
//   V3  V2 + __syncthreads() every 4 sub-tiles
//   V4  V3 + global->register->LDS staging of a fresh 32 KiB tile every 4 sub-tiles (one tile in flight)
//   V5  V0 + the softmax VALU on the SAME sub-tile's accumulators right after the chain (no pipelining)
// Reported: ns per sub-tile per wave (8 MFMAs = 8 * 32 cycles = 116 ns at 2.2 GHz when a wave has the pipe alone,
// 233 ns when two waves share it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITER 1024

__device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void softmax16(const f32x16& ap, float& m, float& z, float c) {
    float tm = ap[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, ap[r]);
    const float mn = fmaxf(m, tm), off = -mn * c;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        s0 += __builtin_amdgcn_exp2f(fmaf(ap[r], c, off));
        s1 += __builtin_amdgcn_exp2f(fmaf(ap[r + 1], c, off));
    }
    z = z * __builtin_amdgcn_exp2f(fmaf(m, c, off)) + s0 + s1;
    m = mn;
}

template <int V>
__global__ __launch_bounds__(512, 2) void k(float* out, const uint4* in, const char* kglob, size_t kbytes) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 32768];
    const uint32_t lane = threadIdx.x & 63, n = lane & 31, kg = lane >> 5;
    uint4 qf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] = in[(lane + i * 7) & 127];
    for (int i = threadIdx.x; i < 4096; i += 512) reinterpret_cast<uint4*>(lds)[i] = in[i & 127];
    __syncthreads();
    f32x16 accA, accB;
    for (int i = 0; i < 16; ++i) { accA[i] = 0.f; accB[i] = 0.f; }
    float m = -1e30f, z = 0.f;
    const float c = 0.1275f;
    const char* src = kglob + (size_t)blockIdx.x * 65536;
    uint4 st[4];
    auto frag = [&](const unsigned char* buf, int sub, int ks) {
        const uint32_t row = sub * 32 + n;
        return *reinterpret_cast<const uint4*>(buf + row * 256 + (((ks * 2 + kg) ^ (row & 15)) << 4));
    };
    for (int it = 0; it < ITER; ++it) {
        const unsigned char* buf = lds + ((it >> 2) & 1) * 32768;
        const int sub = it & 3;
        if (V == 4 && sub == 0) {
            const size_t off = ((size_t)(it >> 2) * 32768 + threadIdx.x * 16) % (kbytes - 65536);
#pragma unroll
            for (int i = 0; i < 4; ++i) st[i] = *reinterpret_cast<const uint4*>(src + off + i * 8192);
        }
        uint4 kf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[ks] = (V >= 2 && V != 5) ? frag(buf, sub, ks) : qf[(ks + 3) & 7];
        f32x16& ac = (it & 1) ? accB : accA;
        f32x16& ap = (it & 1) ? accA : accB;
#pragma unroll
        for (int i = 0; i < 16; ++i) ac[i] = 0.f;
        if (V == 0) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) ac = mma(kf[ks], qf[ks], ac);
            m += ac[0];
        } else if (V == 5) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) ac = mma(kf[ks], qf[ks], ac);
            softmax16(ac, m, z, c);
        } else {
            float tm = ap[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tm = fmaxf(tm, ap[r]);
            const float mn = fmaxf(m, tm), off = -mn * c;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                ac = mma(kf[ks], qf[ks], ac);
                s0 += __builtin_amdgcn_exp2f(fmaf(ap[2 * ks], c, off));
                s1 += __builtin_amdgcn_exp2f(fmaf(ap[2 * ks + 1], c, off));
            }
            z = z * __builtin_amdgcn_exp2f(fmaf(m, c, off)) + s0 + s1;
            m = mn;
        }
        if (V >= 3 && V != 5 && sub == 3) {
            if (V == 4) {
                unsigned char* nb = lds + (((it >> 2) + 1) & 1) * 32768;
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(nb + threadIdx.x * 16 + i * 8192) = st[i];
            }
            __syncthreads();
        }
    }
    float r = m + z;
    for (int i = 0; i < 16; ++i) r += accA[i] + accB[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int V> float run(float* out, const uint4* in, const char* kg, size_t kb) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<V><<<256, 512>>>(out, in, kg, kb); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<V><<<256, 512>>>(out, in, kg, kb); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / ITER;
}
int main() {
    float* out; uint4* in; char* kg; const size_t kb = (size_t)512 << 20;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 128 * sizeof(uint4)); (void)hipMalloc(&kg, kb);
    std::vector<uint32_t> h(512, 0x3c003c00u); (void)hipMemcpy(in, h.data(), 2048, hipMemcpyHostToDevice);
    (void)hipMemset(kg, 0x3c, kb);
    printf("ns per 32-key sub-tile per wave (8 waves per CU, 2 per SIMD); matrix-pipe floor = 233 ns\n");
    printf("V0 mfma only                  %.0f\n", run<0>(out, in, kg, kb));
    printf("V5 mfma then dependent softmax %.0f\n", run<5>(out, in, kg, kb));
    printf("V1 pipelined softmax           %.0f\n", run<1>(out, in, kg, kb));
    printf("V2 + operands from LDS         %.0f\n", run<2>(out, in, kg, kb));
    printf("V3 + barrier per 4 sub-tiles   %.0f\n", run<3>(out, in, kg, kb));
    printf("V4 + global->LDS staging       %.0f\n", run<4>(out, in, kg, kb));
    return 0;
}
