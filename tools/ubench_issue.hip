// Micro-benchmark (asm-level, fixed registers): how do v_mfma_f32_32x32x16_bf16 and VALU / transcendental
// instructions share ONE gfx950 SIMD -- inside one wave's in-order stream and between two co-resident waves?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
// Every variant is a loop of ITER iterations around one asm block; reported per iteration: shader cycles
// (s_memtime, average over waves), wall ns, and the cycles per MFMA of that body.  Registers are fixed:
//   a = v[8:11], b = v[12:15]; accumulators A..D = v[16:31], v[32:47], v[48:63], v[64:79];
//   VALU chains v80..v95 (16 independent), constants v96 (scale), v97 (offset)
// Variants answer: (1) back-to-back MFMAs on one / several accumulators; (2) independent VALU between MFMAs of the
// SAME accumulator (chain) vs DIFFERENT accumulators; (3) how many VALU / exp fit under one MFMA; (4) one wave
// running MFMAs beside a partner wave running VALU on the same SIMD; (5) 16x16x32 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define ITER 4000

#define MFA "v_mfma_f32_32x32x16_bf16 v[16:31], v[8:11], v[12:15], v[16:31]\n"
#define MFB "v_mfma_f32_32x32x16_bf16 v[32:47], v[8:11], v[12:15], v[32:47]\n"
#define MFC "v_mfma_f32_32x32x16_bf16 v[48:63], v[8:11], v[12:15], v[48:63]\n"
#define MFD "v_mfma_f32_32x32x16_bf16 v[64:79], v[8:11], v[12:15], v[64:79]\n"
// 16x16x32: 4 accumulator registers each
#define SFA "v_mfma_f32_16x16x32_bf16 v[16:19], v[8:11], v[12:15], v[16:19]\n"
#define SFB "v_mfma_f32_16x16x32_bf16 v[20:23], v[8:11], v[12:15], v[20:23]\n"
#define SFC "v_mfma_f32_16x16x32_bf16 v[24:27], v[8:11], v[12:15], v[24:27]\n"
#define SFD "v_mfma_f32_16x16x32_bf16 v[28:31], v[8:11], v[12:15], v[28:31]\n"
// independent VALU: chain register r (80..95) is touched once per group of 16 -> never latency-bound
#define FMA(r) "v_fma_f32 v" #r ", v96, v" #r ", v97\n"
#define EXP(r) "v_exp_f32 v" #r ", v" #r "\n"
#define ADD(r) "v_add_f32 v" #r ", v97, v" #r "\n"
#define MAX3(r) "v_max3_f32 v" #r ", v" #r ", v96, v97\n"
#define V6a FMA(80) FMA(81) FMA(82) FMA(83) FMA(84) FMA(85)
#define V6b FMA(86) FMA(87) FMA(88) FMA(89) FMA(90) FMA(91)
#define V4c FMA(92) FMA(93) FMA(94) FMA(95)
#define X6a FMA(80) FMA(81) EXP(82) EXP(83) ADD(84) ADD(85)
#define X6b FMA(86) FMA(87) EXP(88) EXP(89) ADD(90) ADD(91)
#define X3a FMA(80) EXP(82) ADD(84)
#define X3b FMA(86) EXP(88) ADD(90)
#define E2a EXP(82) EXP(83)
#define E2b EXP(88) EXP(89)

#define CLOB                                                                                                              \
    "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", \
        "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40",   \
        "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56",   \
        "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",   \
        "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88",   \
        "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97"

enum {
    B_MF1 = 0,      // 8 MFMAs, one accumulator (dependent chain, back to back)
    B_MF2,          // 8 MFMAs alternating two accumulators
    B_MF4,          // 8 MFMAs over four accumulators
    B_V48,          // 48 independent v_fma
    B_X48,          // 16 x (fma, exp, add)
    B_E32,          // 32 x exp
    B_DEP_V6,       // 8 x (MFMA A, 6 fma)
    B_ALT2_V6,      // 4 x (MFMA A, 6 fma, MFMA B, 6 fma)
    B_ALT4_V6,      // 2 x (A 6 B 6 C 6 D 6)
    B_DEP_X6,       // 8 x (MFMA A, 2 fma 2 exp 2 add)
    B_ALT2_X6,
    B_ALT4_X6,
    B_ALT2_V12,     // 12 fma per gap
    B_ALT2_V10,
    B_ALT2_V16,
    B_ALT2_X9,      // 3 fma 3 exp 3 add per gap
    B_ALT2_E2,      // 2 exp per gap
    B_ALT2_E4,      // 4 exp per gap
    B_ALT2_V2,
    B_ALT2_V4,
    B_DEP_V1,       // one VALU between same-accumulator MFMAs (the cliff)
    B_DEP_V2,
    B_ALT2_V1,
    B_SF1,          // 16 x 16x16x32 MFMA, one accumulator
    B_SF4,          // 16 x 16x16x32 over four accumulators
    B_SF4_V3,       // 16 x (16x16x32 MFMA, 3 fma), four accumulators
    B_SF4_X3,
    B_PAIR2_V6,     // A A' pattern: 2 MFMAs back to back on A, 12 VALU, 2 on B, 12 VALU
    B_COUNT
};

static const char* names[B_COUNT] = {
    "8 MFMA, 1 acc (dependent, back to back)", "8 MFMA, 2 accs alternating", "8 MFMA, 4 accs", "48 v_fma (independent)",
    "16 x (fma, exp, add)", "32 x v_exp", "8 x (MFMA A + 6 fma)  [same acc]", "4 x (A+6fma, B+6fma)", "2 x (A,B,C,D each +6fma)",
    "8 x (MFMA A + 2fma 2exp 2add) [same acc]", "4 x (A+x6, B+x6)", "2 x (A,B,C,D each +x6)", "4 x (A+12fma, B+12fma)",
    "4 x (A+10fma, B+10fma)", "4 x (A+16fma, B+16fma)", "4 x (A+x9, B+x9)", "4 x (A+2exp, B+2exp)", "4 x (A+4exp, B+4exp)",
    "4 x (A+2fma, B+2fma)", "4 x (A+4fma, B+4fma)", "8 x (MFMA A + 1 fma) [same acc]", "8 x (MFMA A + 2 fma) [same acc]",
    "4 x (A+1fma, B+1fma)", "16 x MFMA16x16x32, 1 acc", "16 x MFMA16x16x32, 4 accs", "16 x (MFMA16 + 3 fma), 4 accs",
    "16 x (MFMA16 + fma exp add), 4 accs", "2 x (A,A,12fma,B,B,12fma)"};
static const int n_mfma[B_COUNT] = {8, 8, 8, 0, 0, 0, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 16, 16, 16, 16, 8};

template <int BODY> __device__ __forceinline__ void body() {
    if (BODY == B_MF1) asm volatile(MFA MFA MFA MFA MFA MFA MFA MFA ::: CLOB);
    if (BODY == B_MF2) asm volatile(MFA MFB MFA MFB MFA MFB MFA MFB ::: CLOB);
    if (BODY == B_MF4) asm volatile(MFA MFB MFC MFD MFA MFB MFC MFD ::: CLOB);
    if (BODY == B_V48) asm volatile(V6a V6b V4c V6a V6b V4c V6a V6b V4c ::: CLOB);
    if (BODY == B_X48) asm volatile(X6a X6b X6a X6b X6a X6b X6a X6b ::: CLOB);
    if (BODY == B_E32) asm volatile(E2a E2b E2a E2b E2a E2b E2a E2b E2a E2b E2a E2b E2a E2b E2a E2b ::: CLOB);
    if (BODY == B_DEP_V6) asm volatile(MFA V6a MFA V6b MFA V6a MFA V6b MFA V6a MFA V6b MFA V6a MFA V6b ::: CLOB);
    if (BODY == B_ALT2_V6) asm volatile(MFA V6a MFB V6b MFA V6a MFB V6b MFA V6a MFB V6b MFA V6a MFB V6b ::: CLOB);
    if (BODY == B_ALT4_V6) asm volatile(MFA V6a MFB V6b MFC V6a MFD V6b MFA V6a MFB V6b MFC V6a MFD V6b ::: CLOB);
    if (BODY == B_DEP_X6) asm volatile(MFA X6a MFA X6b MFA X6a MFA X6b MFA X6a MFA X6b MFA X6a MFA X6b ::: CLOB);
    if (BODY == B_ALT2_X6) asm volatile(MFA X6a MFB X6b MFA X6a MFB X6b MFA X6a MFB X6b MFA X6a MFB X6b ::: CLOB);
    if (BODY == B_ALT4_X6) asm volatile(MFA X6a MFB X6b MFC X6a MFD X6b MFA X6a MFB X6b MFC X6a MFD X6b ::: CLOB);
    if (BODY == B_ALT2_V12) asm volatile(MFA V6a V6b MFB V6a V6b MFA V6a V6b MFB V6a V6b MFA V6a V6b MFB V6a V6b MFA V6a V6b MFB V6a V6b ::: CLOB);
    if (BODY == B_ALT2_V10) asm volatile(MFA V6a V4c MFB V6b V4c MFA V6a V4c MFB V6b V4c MFA V6a V4c MFB V6b V4c MFA V6a V4c MFB V6b V4c ::: CLOB);
    if (BODY == B_ALT2_V16) asm volatile(MFA V6a V6b V4c MFB V6a V6b V4c MFA V6a V6b V4c MFB V6a V6b V4c MFA V6a V6b V4c MFB V6a V6b V4c MFA V6a V6b V4c MFB V6a V6b V4c ::: CLOB);
    if (BODY == B_ALT2_X9) asm volatile(MFA X6a X3b MFB X6a X3b MFA X6a X3b MFB X6a X3b MFA X6a X3b MFB X6a X3b MFA X6a X3b MFB X6a X3b ::: CLOB);
    if (BODY == B_ALT2_E2) asm volatile(MFA E2a MFB E2b MFA E2a MFB E2b MFA E2a MFB E2b MFA E2a MFB E2b ::: CLOB);
    if (BODY == B_ALT2_E4) asm volatile(MFA E2a E2b MFB E2a E2b MFA E2a E2b MFB E2a E2b MFA E2a E2b MFB E2a E2b MFA E2a E2b MFB E2a E2b ::: CLOB);
    if (BODY == B_ALT2_V2) asm volatile(MFA FMA(80) FMA(81) MFB FMA(82) FMA(83) MFA FMA(84) FMA(85) MFB FMA(86) FMA(87) MFA FMA(88) FMA(89) MFB FMA(90) FMA(91) MFA FMA(92) FMA(93) MFB FMA(94) FMA(95) ::: CLOB);
    if (BODY == B_ALT2_V4) asm volatile(MFA V4c MFB FMA(80) FMA(81) FMA(82) FMA(83) MFA FMA(84) FMA(85) FMA(86) FMA(87) MFB FMA(88) FMA(89) FMA(90) FMA(91) MFA V4c MFB FMA(80) FMA(81) FMA(82) FMA(83) MFA FMA(84) FMA(85) FMA(86) FMA(87) MFB FMA(88) FMA(89) FMA(90) FMA(91) ::: CLOB);
    if (BODY == B_DEP_V1) asm volatile(MFA FMA(80) MFA FMA(81) MFA FMA(82) MFA FMA(83) MFA FMA(84) MFA FMA(85) MFA FMA(86) MFA FMA(87) ::: CLOB);
    if (BODY == B_DEP_V2) asm volatile(MFA FMA(80) FMA(88) MFA FMA(81) FMA(89) MFA FMA(82) FMA(90) MFA FMA(83) FMA(91) MFA FMA(84) FMA(92) MFA FMA(85) FMA(93) MFA FMA(86) FMA(94) MFA FMA(87) FMA(95) ::: CLOB);
    if (BODY == B_ALT2_V1) asm volatile(MFA FMA(80) MFB FMA(81) MFA FMA(82) MFB FMA(83) MFA FMA(84) MFB FMA(85) MFA FMA(86) MFB FMA(87) ::: CLOB);
    if (BODY == B_SF1) asm volatile(SFA SFA SFA SFA SFA SFA SFA SFA SFA SFA SFA SFA SFA SFA SFA SFA ::: CLOB);
    if (BODY == B_SF4) asm volatile(SFA SFB SFC SFD SFA SFB SFC SFD SFA SFB SFC SFD SFA SFB SFC SFD ::: CLOB);
    if (BODY == B_SF4_V3)
        asm volatile(SFA FMA(80) FMA(81) FMA(82) SFB FMA(83) FMA(84) FMA(85) SFC FMA(86) FMA(87) FMA(88) SFD FMA(89) FMA(90) FMA(91)
                     SFA FMA(92) FMA(93) FMA(94) SFB FMA(95) FMA(80) FMA(81) SFC FMA(82) FMA(83) FMA(84) SFD FMA(85) FMA(86) FMA(87)
                     SFA FMA(88) FMA(89) FMA(90) SFB FMA(91) FMA(92) FMA(93) SFC FMA(94) FMA(95) FMA(80) SFD FMA(81) FMA(82) FMA(83)
                     SFA FMA(84) FMA(85) FMA(86) SFB FMA(87) FMA(88) FMA(89) SFC FMA(90) FMA(91) FMA(92) SFD FMA(93) FMA(94) FMA(95) ::: CLOB);
    if (BODY == B_SF4_X3)
        asm volatile(SFA X3a SFB X3b SFC X3a SFD X3b SFA X3a SFB X3b SFC X3a SFD X3b SFA X3a SFB X3b SFC X3a SFD X3b SFA X3a SFB X3b SFC X3a SFD X3b ::: CLOB);
    if (BODY == B_PAIR2_V6) asm volatile(MFA MFA V6a V6b MFB MFB V6a V6b MFA MFA V6a V6b MFB MFB V6a V6b ::: CLOB);
}

// ROLE: 0 every wave runs BODY; 1 the first wave of each SIMD (waves 0-3) runs BODY, the second (waves 4-7) runs BODY2
template <int BODY, int BODY2>
__global__ void k(unsigned long long* cyc, float* sink, const uint32_t* in, int split) {
    // operands: random bf16 pairs (power draw of real data) or constants, per `in`
    asm volatile(
        "global_load_dwordx4 v[8:11], %0, off\n"
        "global_load_dwordx4 v[12:15], %0, off offset:16\n"
        "s_waitcnt vmcnt(0)\n"
        "v_mov_b32 v96, 0.5\n v_mov_b32 v97, 1.0\n"
        :: "v"(in + (threadIdx.x & 63) * 8) : CLOB, "memory");
#define ZERO4(a, b, c, d) "v_mov_b32 v" #a ", 0\n v_mov_b32 v" #b ", 0\n v_mov_b32 v" #c ", 0\n v_mov_b32 v" #d ", 0\n"
    asm volatile(ZERO4(16, 17, 18, 19) ZERO4(20, 21, 22, 23) ZERO4(24, 25, 26, 27) ZERO4(28, 29, 30, 31) ZERO4(32, 33, 34, 35) ZERO4(36, 37, 38, 39)
                 ZERO4(40, 41, 42, 43) ZERO4(44, 45, 46, 47) ZERO4(48, 49, 50, 51) ZERO4(52, 53, 54, 55) ZERO4(56, 57, 58, 59) ZERO4(60, 61, 62, 63)
                 ZERO4(64, 65, 66, 67) ZERO4(68, 69, 70, 71) ZERO4(72, 73, 74, 75) ZERO4(76, 77, 78, 79) ZERO4(80, 81, 82, 83) ZERO4(84, 85, 86, 87)
                 ZERO4(88, 89, 90, 91) ZERO4(92, 93, 94, 95) ::: CLOB);
    __syncthreads();
    const bool second = split && (threadIdx.x >> 6) >= 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (!second) {
        for (int it = 0; it < ITER; ++it) body<BODY>();
    } else {
        for (int it = 0; it < ITER; ++it) body<BODY2>();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r;
    asm volatile("s_nop 15\n s_nop 15\n v_add_f32 %0, v16, v80\n v_add_f32 %0, %0, v32\n v_add_f32 %0, %0, v48\n v_add_f32 %0, %0, v64" : "=v"(r)::CLOB);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (r == 123.456f) sink[0] = r;
}

struct Res { double cyc_a, cyc_b, ns; };

template <int BODY, int BODY2>
Res run(int threads, int blocks, int split, unsigned long long* d_cyc, float* sink, const uint32_t* in) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<BODY, BODY2><<<blocks, threads>>>(d_cyc, sink, in, split);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<BODY, BODY2><<<blocks, threads>>>(d_cyc, sink, in, split);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int wpb = threads / 64;
    std::vector<unsigned long long> h((size_t)blocks * wpb);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0; int na = 0, nb = 0;
    for (int i = 0; i < blocks; ++i)
        for (int w = 0; w < wpb; ++w) {
            if (split && w >= 4) { b += (double)h[(size_t)i * wpb + w]; ++nb; }
            else { a += (double)h[(size_t)i * wpb + w]; ++na; }
        }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return {a / na / ITER, nb ? b / nb / ITER : 0.0, ms * 1e6 / ITER};
}

template <int BODY> void report(const char* tag, int threads, int blocks, unsigned long long* d_cyc, float* sink, const uint32_t* in) {
    const Res r = run<BODY, BODY>(threads, blocks, 0, d_cyc, sink, in);
    printf("%-7s %-46s %8.1f cyc/iter  %7.1f ns/iter", tag, names[BODY], r.cyc_a, r.ns);
    if (n_mfma[BODY]) printf("   %6.1f cyc/MFMA", r.cyc_a / n_mfma[BODY]);
    printf("   (%.0f MHz)\n", r.cyc_a / r.ns * 1e3);
}
template <int B1, int B2> void report_split(const char* tag, int blocks, unsigned long long* d_cyc, float* sink, const uint32_t* in) {
    const Res r = run<B1, B2>(512, blocks, 1, d_cyc, sink, in);
    printf("%-7s first wave/SIMD: %-38s %8.1f cyc/iter | second: %-30s %8.1f cyc/iter | %7.1f ns/iter\n", tag, names[B1], r.cyc_a, names[B2],
           r.cyc_b, r.ns);
}

template <int B> struct All {
    static void go(const char* tag, int threads, int blocks, unsigned long long* c, float* s, const uint32_t* in) {
        report<B>(tag, threads, blocks, c, s, in);
        All<B + 1>::go(tag, threads, blocks, c, s, in);
    }
};
template <> struct All<B_COUNT> {
    static void go(const char*, int, int, unsigned long long*, float*, const uint32_t*) {}
};

int main(int argc, char** argv) {
    unsigned long long* d_cyc; float* sink; uint32_t* in_rand; uint32_t* in_const;
    hipMalloc(&d_cyc, 4096 * 8 * 8);
    hipMalloc(&sink, 64);
    hipMalloc(&in_rand, 64 * 8 * 4);
    hipMalloc(&in_const, 64 * 8 * 4);
    std::vector<uint32_t> h(64 * 8);
    srand(1);
    for (auto& x : h) {  // two random bf16 in [-2, 2)
        auto bf = [] { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; uint32_t u; memcpy(&u, &f, 4); return u >> 16; };
        x = bf() | (bf() << 16);
    }
    hipMemcpy(in_rand, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (auto& x : h) x = 0x3c003c00u;
    hipMemcpy(in_const, h.data(), h.size() * 4, hipMemcpyHostToDevice);

    printf("== one wave per SIMD (256 threads), 8 workgroups (no power limit), constant operands\n");
    All<0>::go("1w/8", 256, 8, d_cyc, sink, in_const);
    printf("== two waves per SIMD (512 threads), 8 workgroups, constant operands\n");
    All<0>::go("2w/8", 512, 8, d_cyc, sink, in_const);
    printf("== two waves per SIMD, 256 workgroups (whole chip), RANDOM operands\n");
    All<0>::go("2w/256r", 512, 256, d_cyc, sink, in_rand);
    printf("== one wave MFMA, partner wave VALU on the same SIMD (512 threads, 8 workgroups)\n");
    report_split<B_MF1, B_V48>("split", 8, d_cyc, sink, in_const);
    report_split<B_MF2, B_V48>("split", 8, d_cyc, sink, in_const);
    report_split<B_MF2, B_X48>("split", 8, d_cyc, sink, in_const);
    report_split<B_MF2, B_E32>("split", 8, d_cyc, sink, in_const);
    report_split<B_MF1, B_MF1>("split", 8, d_cyc, sink, in_const);
    report_split<B_ALT2_X6, B_ALT2_X6>("split", 8, d_cyc, sink, in_const);
    report_split<B_DEP_X6, B_DEP_X6>("split", 8, d_cyc, sink, in_const);
    return 0;
}
