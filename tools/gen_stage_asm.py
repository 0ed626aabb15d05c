#!/usr/bin/env python3
"""Generator for the hand-scheduled gfx950 inner loops of the SnapKV window-attention passes.

The passes (kvpress_amd/csrc/snapkv_mfma.hip) are bound by how one in-order wave's instruction stream feeds the
SIMD's matrix pipe, VALU / transcendental pipe and LDS at the same time (tools/ubench_issue.hip: an MFMA holds
the matrix pipe for 32 cycles, a wave issues one VALU per ~4.6 cycles, v_exp_f32 costs 8 SIMD cycles, and independent
VALU work runs in the shadow of an MFMA).  hipcc's scheduler does not produce that interleave, so the steady-state tile
loop is emitted here as ONE inline-asm block with fixed registers.

This file emits
  * tools/ubench_stage.hip  (`python tools/gen_stage_asm.py ubench`): the stage in isolation, ingredient by ingredient,
  * kvpress_amd/csrc/snapkv_asm_p1.inc / snapkv_asm_p2.inc (`... kernel`): the production loops included by snapkv_mfma.hip.

Register map (VGPR numbers are fixed inside the asm block and listed as clobbers):
  QF   v[32:63]    Q fragments, 8 k-steps x 4 dwords (B operand in pass 1, A operand in pass 2)
  KF0  v[64:95]    K fragments of the sub-tile being multiplied (8 x ds_read_b128)
  KF1  v[96:127]   K fragments of the next sub-tile (loaded while KF0 is multiplied)
  ACC  v[128:175]  three 16-register accumulators: written by the MFMA chain (W), max-reduced (M), exponentiated (X)
  T    v[176:191]  fma / exp temporaries
  misc v[192:..]   running max m, running sum z, offsets, partial sums
Pass 1 pipeline per 32-key sub-tile s (one "stage"): MFMA chain of s  ||  row maximum of s-1  ||  exp / sum of s-2.
"""
import sys

QF, KF0, KF1 = 32, 64, 96
ACC = [128, 144, 160]
T = 176
M_, Z_, OFFX, RX, OFFM, RM, S0, S1, TMAX, MNEW = 192, 193, 194, 195, 196, 197, 198, 199, 200, 201
LADDR = 202      # 8 per-lane LDS byte offsets (one per k-step), buffer-relative
LADDR2 = 210     # the same + 65536 (third ring buffer: ds offset field is 16 bits)
DMAV = 218       # 4 per-lane global byte offsets (one per request of a tile)
LAST_V = 224


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def mfma(acc, a, b, first, dt="bf16"):
    c = "0" if first else vr(acc, 16)
    return f"v_mfma_f32_32x32x16_{dt} {vr(acc, 16)}, {vr(a, 4)}, {vr(b, 4)}, {c}"


def p1_valu_x(accx, cs="s20"):
    """exp / sum part for the sub-tile in accx: uses OFFX (= -c * m_j) and RX (= 2^(c m_{j-1} - c m_j))."""
    ops = []
    F = lambda i: f"v_fma_f32 v{T + i}, {cs}, v{accx + i}, v{OFFX}"
    E = lambda i: f"v_exp_f32 v{T + i}, v{T + i}"
    A = lambda i: f"v_add_f32 v{S0 if i % 2 == 0 else S1}, v{S0 if i % 2 == 0 else S1}, v{T + i}"
    # z <- z * RX first (independent of the exps), then the partial sums start from the first two exps
    ops.append(f"v_mul_f32 v{Z_}, v{Z_}, v{RX}")
    for b in range(4):
        ops += [F(4 * b + i) for i in range(4)]
        ops += [E(4 * b + i) for i in range(4)]
        if b >= 1:
            for i in range(4):
                j = 4 * (b - 1) + i
                if j < 2:
                    ops.append(f"v_mov_b32 v{S0 if j == 0 else S1}, v{T + j}")
                else:
                    ops.append(A(j))
    ops += [A(12 + i) for i in range(4)]
    ops.append(f"v_add_f32 v{S0}, v{S0}, v{S1}")
    ops.append(f"v_add_f32 v{Z_}, v{Z_}, v{S0}")
    return ops


def p1_valu_m(accm, cs="s20"):
    """row maximum of the sub-tile in accm -> new running max, the offset and rescale factor its exp part will use."""
    ops = [f"v_max3_f32 v{TMAX}, v{accm}, v{accm + 1}, v{accm + 2}"]
    for i in range(3, 15, 2):
        ops.append(f"v_max3_f32 v{TMAX}, v{TMAX}, v{accm + i}, v{accm + i + 1}")
    ops.append(f"v_max3_f32 v{MNEW}, v{M_}, v{TMAX}, v{accm + 15}")
    ops.append(f"v_mul_f32_e64 v{OFFM}, {cs}, -v{MNEW}")
    ops.append(f"v_fma_f32 v{RM}, {cs}, v{M_}, v{OFFM}")
    ops.append(f"v_exp_f32 v{RM}, v{RM}")
    ops.append(f"v_mov_b32 v{M_}, v{MNEW}")
    return ops


def rotate_m_to_x():
    return [f"v_mov_b32 v{OFFX}, v{OFFM}", f"v_mov_b32 v{RX}, v{RM}"]


def spread(slots, ops, start=0, end=None):
    """distribute ops (in order) over slots[start:end] as evenly as possible"""
    end = len(slots) if end is None else end
    n = end - start
    per, rem = divmod(len(ops), n)
    k = 0
    for i in range(n):
        cnt = per + (1 if i < rem else 0)
        slots[start + i] += ops[k:k + cnt]
        k += cnt


def p1_stage(s, opts, lds_imm=None, dma=None):
    """one pass-1 stage (sub-tile index s in the unrolled loop).  Returns asm lines.
    accumulator roles rotate with s: W = ACC[s % 3], M = ACC[(s - 1) % 3], X = ACC[(s - 2) % 3];
    K fragments: multiply KF[s % 2], load KF[(s + 1) % 2]."""
    accw, accm, accx = ACC[s % 3], ACC[(s - 1) % 3], ACC[(s - 2) % 3]
    kfu = KF0 if s % 2 == 0 else KF1
    kfl = KF1 if s % 2 == 0 else KF0
    slots = [[] for _ in range(8)]
    pre = []
    if opts.get("lds"):
        # fragments of the NEXT sub-tile; consumed next stage after s_waitcnt lgkmcnt(0) at its head
        buf, sub = lds_imm
        base = LADDR2 if buf == 2 else LADDR
        imm = (buf % 2) * 32768 + sub * 8192 if buf < 2 else sub * 8192
        reads = [f"ds_read_b128 {vr(kfl + 4 * ks, 4)}, v{base + ks} offset:{imm}" for ks in range(8)]
    else:
        reads = []
    valu = []
    if opts.get("softmax", True):
        x = p1_valu_x(accx)
        m = p1_valu_m(accm)
        # exp part first (its inputs are two stages old), the max chain late (its accumulator finished last stage)
        valu = x[:len(x) // 2] + m[:4] + x[len(x) // 2:] + m[4:] + rotate_m_to_x()
        # NB: m reads RX/OFFX? no: it writes OFFM/RM; x reads OFFX/RX -- rotate only after both are done
    if dma is not None:
        pre += dma
    lines = []
    if opts.get("lds"):
        lines.append("s_waitcnt lgkmcnt(0)")
    lines += pre
    nread = opts.get("reads_per_slot", 2)
    ri = 0
    if opts.get("m16"):
        # 16x16x32 MFMAs: 2 key blocks x 2 row blocks x 4 k-steps; accumulator block (kb, rb) = 4 registers
        slots = [[] for _ in range(16)]
        spread(slots, valu)
        k = 0
        for ks in range(4):
            for kb in range(2):
                for rb in range(2):
                    a = accw + 4 * (kb * 2 + rb)
                    c = "0" if ks == 0 else vr(a, 4)
                    if opts.get("mfma", True):
                        lines.append(f"v_mfma_f32_16x16x32_bf16 {vr(a, 4)}, {vr(kfu + 4 * (kb * 4 + ks), 4)}, {vr(QF + 4 * (rb * 4 + ks), 4)}, {c}")
                    if ri < len(reads) and k % 2 == 0:
                        lines.append(reads[ri])
                        ri += 1
                    lines += slots[k]
                    k += 1
        lines += reads[ri:]
        return [l for l in lines if l]
    spread(slots, valu)
    for k in range(8):
        lines.append(mfma(accw, kfu + 4 * k, QF + 4 * k, k == 0, opts.get("dt", "bf16")) if opts.get("mfma", True) else "")
        # LDS reads go into the first slots (the data is needed at the head of the next stage)
        for _ in range(nread):
            if ri < len(reads):
                lines.append(reads[ri])
                ri += 1
        lines += slots[k]
    lines += reads[ri:]
    return [l for l in lines if l]


def clobbers(lo=32, hi=LAST_V):
    return ", ".join(f'"v{i}"' for i in range(lo, hi))


# ------------------------------------------------------------------------------------------------------------------
# micro-benchmark
# ------------------------------------------------------------------------------------------------------------------
UB_HEAD = r'''// GENERATED by tools/gen_stage_asm.py ubench -- do not edit.
// Pass-1 stage (8 MFMA + online-softmax pieces of the two previous sub-tiles, 3-accumulator pipeline) in isolation.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stage.hip -o tools/ubench_stage && tools/ubench_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define ITER 400
#define CLOB %(clob)s
'''

UB_KERNEL = r'''
__global__ __launch_bounds__(512, 1) void k_%(name)s(unsigned long long* cyc, float* sink, const uint32_t* in, const char* kglob) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x & 63, n = lane & 31, kg = lane >> 5;
    for (uint32_t i = threadIdx.x; i < 98304 / 4; i += 512) reinterpret_cast<uint32_t*>(lds)[i] = in[i & 511];
    __syncthreads();
    const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    uint32_t la[8];
    for (int ks = 0; ks < 8; ++ks) la[ks] = ldsbase + n * 256 + (((ks * 2 + kg) ^ (n & 15)) << 4);   // the kernel's swizzled fragment address
    const uint32_t m0base = __builtin_amdgcn_readfirstlane(ldsbase + (threadIdx.x >> 6) * 1024);
    const char* g = kglob + (size_t)blockIdx.x * ((size_t)400 * 3 * 32768 + 98304);
    const uint32_t voff = threadIdx.x * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(
%(init)s
        "s_mov_b32 s20, 0x3e0296b3\n"   // c = log2(e) / sqrt(128)
        "s_movk_i32 s21, %(iters)d\n"
        "1:\n"
%(body)s
        "s_sub_u32 s21, s21, 1\n"
        "s_cmp_lg_u32 s21, 0\n"
        "s_cbranch_scc1 1b\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "s_nop 15\n"
        :: "v"(in + lane * 8), "v"(voff), "s"(m0base), "s"(g), "v"(la[0]), "v"(la[1]), "v"(la[2]), "v"(la[3]), "v"(la[4]), "v"(la[5]), "v"(la[6]), "v"(la[7])
        : CLOB, "s20", "s21", "s22", "s23", "s24", "s25", "m0", "scc", "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r;
    asm volatile("v_add_f32 %%0, v193, v192" : "=v"(r));
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (r == 123.456f) sink[0] = r;
}
'''


def ub_init():
    L = []
    L.append("global_load_dwordx4 v[32:35], %0, off")
    L.append("global_load_dwordx4 v[36:39], %0, off offset:16")
    L.append("s_waitcnt vmcnt(0)")
    for i in range(40, 128, 4):
        L.append(f"v_mov_b32 v{i}, v{32 + (i % 8)}")
        L.append(f"v_mov_b32 v{i + 1}, v{33 + (i % 7)}")
        L.append(f"v_mov_b32 v{i + 2}, v{32 + (i % 5)}")
        L.append(f"v_mov_b32 v{i + 3}, v{34 + (i % 6)}")
    for i in range(128, LAST_V):
        L.append(f"v_mov_b32 v{i}, 0")
    for ks in range(8):  # LDS fragment addresses (operands %4..%11); second set + 64 KiB for the third ring buffer
        L.append(f"v_mov_b32 v{LADDR + ks}, %{4 + ks}")
        L.append(f"v_add_u32 v{LADDR2 + ks}, 0x10000, %{4 + ks}")
    L.append(f"v_mov_b32 v{M_}, 0xff800000")
    return L


def fmt(lines, indent="        "):
    return "\n".join(f'{indent}"{l}\\n"' for l in lines)


def gen_ubench():
    variants = {
        "mfma_only": dict(softmax=False),
        "valu_only": dict(mfma=False),
        "math": dict(),
        "math_lds": dict(lds=True),
        "math_lds1": dict(lds=True, reads_per_slot=1),
        "math_lds4": dict(lds=True, reads_per_slot=4),
        "math_lds_bar": dict(lds=True, bar=True),
        "math_lds_bar_dma": dict(lds=True, bar=True, dma=True),
        "mfma_lds": dict(softmax=False, lds=True),
        "m16_mfma_only": dict(softmax=False, m16=True),
        "m16_math": dict(m16=True),
        "m16_math_lds": dict(m16=True, lds=True),
        "m16_math_lds_bar": dict(m16=True, lds=True, bar=True),
        "m16_math_lds_bar_dma": dict(m16=True, lds=True, bar=True, dma=True),
        "valu_lds": dict(mfma=False, lds=True),
    }
    out = [UB_HEAD % dict(clob=clobbers())]
    for name, o in variants.items():
        body = []
        # 12 stages = 3 tiles (ring of three buffers), accumulators rotate with period 3, fragments with period 2
        for s in range(12):
            tile, sub = divmod(s, 4)
            nxt_tile, nxt_sub = divmod((s + 1) % 12, 4)
            dma = None
            if o.get("dma"):
                # one request per stage: tile (tile + 2) % 3's sub-block `sub`; M0 = LDS destination of this wave
                dma = [f"s_add_u32 m0, s22, {((tile + 2) % 3) * 32768 + sub * 8192}", "s_nop 0",
                       f"global_load_lds_dwordx4 v{DMAV + sub}, s[24:25]"]
                if sub == 3:
                    dma += ["s_add_u32 s24, s24, 0x8000", "s_addc_u32 s25, s25, 0"]
            body += p1_stage(s, o, lds_imm=(nxt_tile, nxt_sub), dma=dma)
            if sub == 3 and o.get("bar"):
                if o.get("dma"):
                    body.append("s_waitcnt vmcnt(4)")
                body.append("s_barrier")
        init = ub_init()
        if o.get("dma"):
            # s22 = LDS base + 1 KiB * wave (this wave's block inside a sub-block), s[24:25] = global base of this workgroup's stream
            init += ["s_mov_b32 s22, %2", "s_mov_b64 s[24:25], %3"]
            init += [f"v_add_u32 v{DMAV + i}, {i * 8192}, %1" for i in range(4)]
        out.append(UB_KERNEL % dict(name=name, init=fmt(init), body=fmt(body), iters=400 // 1))
    # host
    out.append(r'''
template <typename K> void run(const char* name, K kern, int blocks, const uint32_t* in, const char* kglob, unsigned long long* d_cyc, float* sink) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 512, 98304>>>(d_cyc, sink, in, kglob);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 512, 98304>>>(d_cyc, sink, in, kglob);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0; for (auto x : h) a += (double)x;
    const double stages = 400.0 * 12.0;
    printf("%-22s blocks %3d: %7.1f ns per stage (2 waves/SIMD)   %7.1f s_memtime ticks per stage   kernel %.1f us\n", name, blocks, ms * 1e6 / stages, a / h.size() / stages, ms * 1e3);
}
int main() {
    unsigned long long* d_cyc; float* sink; uint32_t* in_rand; uint32_t* in_const; char* kglob;
    hipMalloc(&d_cyc, 4096 * 8); hipMalloc(&sink, 64); hipMalloc(&in_rand, 512 * 4); hipMalloc(&in_const, 512 * 4);
    const size_t gbytes = (size_t)256 * ((size_t)400 * 3 * 32768 + 98304) + (1u << 20);
    hipMalloc(&kglob, gbytes);
    hipMemset(kglob, 0x3c, gbytes);
    std::vector<uint32_t> h(512);
    srand(1);
    for (auto& x : h) { auto bf = [] { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; uint32_t u; memcpy(&u, &f, 4); return u >> 16; }; x = bf() | (bf() << 16); }
    hipMemcpy(in_rand, h.data(), 2048, hipMemcpyHostToDevice);
    for (auto& x : h) x = 0x3c003c00u;
    hipMemcpy(in_const, h.data(), 2048, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        const int blocks = mode ? 256 : 8;
        const uint32_t* in = mode ? in_rand : in_const;
        printf("== %s\n", mode ? "256 workgroups, random operands" : "8 workgroups, constant operands");
''')
    for name in variants:
        out.append(f'        run("{name}", k_{name}, blocks, in, kglob, d_cyc, sink);\n')
    out.append("    }\n    return 0;\n}\n")
    return "".join(out)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "ubench"
    if what == "ubench":
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_stage.hip")
        open(path, "w").write(gen_ubench())
        print(path)
