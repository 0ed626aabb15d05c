#!/usr/bin/env python3
"""Generator for the hand-scheduled gfx950 inner loops of the SnapKV window-attention passes.

The passes (kvpress_amd/csrc/snapkv_mfma.hip) are bound by how one in-order wave's instruction stream feeds the
SIMD's matrix pipe, VALU / transcendental pipe and LDS at the same time (tools/ubench_issue.hip: an MFMA holds
the matrix pipe for 32 cycles, a wave issues one VALU per ~4.6 cycles, v_exp_f32 costs 8 SIMD cycles, and independent
VALU work runs in the shadow of an MFMA).  hipcc's scheduler does not produce that interleave, so the steady-state tile
loop is emitted here as ONE inline-asm block with fixed registers.

This file emits
  * tools/ubench_stage.hip  (`python tools/gen_stage_asm.py ubench`): the stage in isolation, ingredient by ingredient
    (first part of this file, names prefixed UB_ / ub_: the micro-benchmark's own fixed register map),
  * kvpress_amd/csrc/snapkv_asm.inc (`python tools/gen_stage_asm.py kernel`): the production loops of both passes, included by
    snapkv_mfma.hip (second part: class Cfg holds the register map and ring geometry of a pass).
    GEN_ABL=nodma,nobar,novalu,nomfma,nolds / GEN_P1_NBUF=4 / GEN_PK=fma,add produce the lab variants of DESIGN.md section 6
    (tools/build_variants.sh); tests/test_capi_symbols.py checks that the committed .inc is what the default settings emit.

Register map (VGPR numbers are fixed inside the asm block and listed as clobbers):
  QF   v[32:63]    Q fragments, 8 k-steps x 4 dwords (B operand in pass 1, A operand in pass 2)
  KF0  v[64:95]    K fragments of the sub-tile being multiplied (8 x ds_read_b128)
  KF1  v[96:127]   K fragments of the next sub-tile (loaded while KF0 is multiplied)
  ACC  v[128:175]  three 16-register accumulators: written by the MFMA chain (W), max-reduced (M), exponentiated (X)
  T    v[176:191]  fma / exp temporaries
  misc v[192:..]   running max m, running sum z, offsets, partial sums, LDS / global addresses
Pass 1 pipeline per 32-key sub-tile s (one "stage"): MFMA chain of s  ||  row maximum of s-1  ||  exp / sum of s-2.
Pass 2: MFMA chain of s  ||  exp / column sums of s-2 (-> LDS slots, flushed one tile later).
"""
import sys

QF, KF0, KF1 = 32, 64, 96
ACC = [128, 144, 160]
UB_T = 176
UB_M_, UB_Z_, UB_OFFX, UB_RX, UB_OFFM, UB_RM, UB_S0, UB_S1, UB_TMAX, UB_MNEW = 192, 193, 194, 195, 196, 197, 198, 199, 200, 201
LADDR = 202      # 8 per-lane LDS byte offsets (one per k-step), buffer-relative
LADDR2 = 210     # the same + 65536 (third ring buffer: ds offset field is 16 bits)
DMAV = 218       # 4 per-lane global byte offsets (one per request of a tile)
LAST_V = 224


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def mfma(acc, a, b, first, dt="bf16"):
    c = "0" if first else vr(acc, 16)
    return f"v_mfma_f32_32x32x16_{dt} {vr(acc, 16)}, {vr(a, 4)}, {vr(b, 4)}, {c}"


def ub_p1_valu_x(accx, cs="s20"):
    """exp / sum part for the sub-tile in accx: uses UB_OFFX (= -c * m_j) and UB_RX (= 2^(c m_{j-1} - c m_j))."""
    ops = []
    F = lambda i: f"v_fma_f32 v{UB_T + i}, {cs}, v{accx + i}, v{UB_OFFX}"
    E = lambda i: f"v_exp_f32 v{UB_T + i}, v{UB_T + i}"
    A = lambda i: f"v_add_f32 v{UB_S0 if i % 2 == 0 else UB_S1}, v{UB_S0 if i % 2 == 0 else UB_S1}, v{UB_T + i}"
    # z <- z * UB_RX first (independent of the exps), then the partial sums start from the first two exps
    ops.append(f"v_mul_f32 v{UB_Z_}, v{UB_Z_}, v{UB_RX}")
    for b in range(4):
        ops += [F(4 * b + i) for i in range(4)]
        ops += [E(4 * b + i) for i in range(4)]
        if b >= 1:
            for i in range(4):
                j = 4 * (b - 1) + i
                if j < 2:
                    ops.append(f"v_mov_b32 v{UB_S0 if j == 0 else UB_S1}, v{UB_T + j}")
                else:
                    ops.append(A(j))
    ops += [A(12 + i) for i in range(4)]
    ops.append(f"v_add_f32 v{UB_S0}, v{UB_S0}, v{UB_S1}")
    ops.append(f"v_add_f32 v{UB_Z_}, v{UB_Z_}, v{UB_S0}")
    return ops


def ub_p1_valu_m(accm, cs="s20"):
    """row maximum of the sub-tile in accm -> new running max, the offset and rescale factor its exp part will use."""
    ops = [f"v_max3_f32 v{UB_TMAX}, v{accm}, v{accm + 1}, v{accm + 2}"]
    for i in range(3, 15, 2):
        ops.append(f"v_max3_f32 v{UB_TMAX}, v{UB_TMAX}, v{accm + i}, v{accm + i + 1}")
    ops.append(f"v_max3_f32 v{UB_MNEW}, v{UB_M_}, v{UB_TMAX}, v{accm + 15}")
    ops.append(f"v_mul_f32_e64 v{UB_OFFM}, {cs}, -v{UB_MNEW}")
    ops.append(f"v_fma_f32 v{UB_RM}, {cs}, v{UB_M_}, v{UB_OFFM}")
    ops.append(f"v_exp_f32 v{UB_RM}, v{UB_RM}")
    ops.append(f"v_mov_b32 v{UB_M_}, v{UB_MNEW}")
    return ops


def ub_rotate_m_to_x():
    return [f"v_mov_b32 v{UB_OFFX}, v{UB_OFFM}", f"v_mov_b32 v{UB_RX}, v{UB_RM}"]


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
        x = ub_p1_valu_x(accx)
        m = ub_p1_valu_m(accm)
        # exp part first (its inputs are two stages old), the max chain late (its accumulator finished last stage)
        valu = x[:len(x) // 2] + m[:4] + x[len(x) // 2:] + m[4:] + ub_rotate_m_to_x()
        # NB: m reads UB_RX/UB_OFFX? no: it writes UB_OFFM/UB_RM; x reads UB_OFFX/UB_RX -- rotate only after both are done
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


# ---- one wave per SIMD (4-wave workgroup): a wave owns a whole q-head (both 32-row halves of the window) -------------------------
# K fragments in AGPRs a[0:31] / a[32:63] (MFMA A operands may be AGPRs; ds_read may target them), the second half's accumulators in
# the VGPRs the fragments freed.  Per stage: 16 MFMAs (two chains sharing their K fragments), the softmax pieces of both halves, 8
# fragment reads, two LDS-DMA pieces.
ACCB = [64, 80, 96]


def w1_stage(s, opts, lds_imm=None, dma=None):
    A = (ACC[s % 3], ACC[(s - 1) % 3], ACC[(s - 2) % 3])
    B = (ACCB[s % 3], ACCB[(s - 1) % 3], ACCB[(s - 2) % 3])
    kfu, kfl = (0, 32) if s % 2 == 0 else (32, 0)
    lines = []
    reads = []
    if opts.get("lds"):
        buf, sub = lds_imm
        base = LADDR2 if buf == 2 else LADDR
        imm = (buf % 2) * 32768 + sub * 8192 if buf < 2 else sub * 8192
        reads = [f"ds_read_b128 a[{kfl + 4 * ks}:{kfl + 4 * ks + 3}], v{base + ks} offset:{imm}" for ks in range(8)]
        lines.append("s_waitcnt lgkmcnt(0)")
    if dma is not None:
        lines += dma
    valu = []
    if opts.get("softmax", True):
        for acc in (A, B):
            x, m = ub_p1_valu_x(acc[2]), ub_p1_valu_m(acc[1])
            valu += x[:len(x) // 2] + m[:4] + x[len(x) // 2:] + m[4:] + ub_rotate_m_to_x()
    slots = [[] for _ in range(16)]
    spread(slots, valu)
    ri = 0
    for k in range(8):
        for h, acc in enumerate((A, B)):
            if opts.get("mfma", True):
                c = "0" if k == 0 else vr(acc[0], 16)
                lines.append(f"v_mfma_f32_32x32x16_bf16 {vr(acc[0], 16)}, a[{kfu + 4 * k}:{kfu + 4 * k + 3}], {vr(QF + 4 * k, 4)}, {c}")
            if h == 0 and ri < len(reads):
                lines.append(reads[ri]); ri += 1
            lines += slots[2 * k + h]
    lines += reads[ri:]
    return [l for l in lines if l]


def mfma_reuse_body(kind, dt="bf16"):
    """96 MFMAs (= 12 stages of 8) with a chosen operand-reuse pattern between CONSECUTIVE matrix instructions (lab: does the matrix
    pipe draw less when an operand repeats?  the chip is power-limited in these loops, so time ~ energy).
      afix / bfix / abfix   every instruction names the same A / B / both registers (bound of the effect)
      bpair / bquad         k-step outer, 2 / 4 sub-tiles inner: B (the Q fragment of pass 1) repeats 2 / 4 times, A always differs
      apair / aquad         the same with the roles swapped (pass 2: A = Q fragment)
    accumulators: v128.. (4 x 16; the softmax temporaries are unused here)."""
    accs = [128, 144, 160, 176]
    kregs = [KF0, KF1]
    L = []
    if kind in ("afix", "bfix", "abfix"):
        for s_ in range(12):
            kfu = kregs[s_ % 2]
            for k in range(8):
                a = kfu if kind in ("afix", "abfix") else kfu + 4 * k
                b = QF if kind in ("bfix", "abfix") else QF + 4 * k
                L.append(mfma(accs[s_ % 3], a, b, k == 0, dt))
        return L
    nsub = 2 if kind.endswith("pair") else 4
    shared_b = kind.startswith("b")
    for g in range(12 // nsub):
        for k in range(8):
            for sub in range(nsub):
                # the streaming operand: a different register set per sub-tile (values differ: ub_init permutes 8 random dwords)
                stream = kregs[sub % 2] + 4 * ((k + 3 * (sub // 2)) % 8)
                shared = QF + 4 * k
                a, b = (stream, shared) if shared_b else (shared, stream)
                L.append(mfma(accs[sub], a, b, k == 0, dt))
    return L


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
#include <sys/time.h>
#include <vector>
#define ITER 400
#define CLOB %(clob)s
'''

UB_KERNEL = r'''
__global__ __launch_bounds__(%(threads)d, 1) void k_%(name)s(unsigned long long* cyc, float* sink, const uint32_t* in, const char* kglob) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x & 63, n = lane & 31, kg = lane >> 5;
    for (uint32_t i = threadIdx.x; i < 98304 / 4; i += %(threads)d) reinterpret_cast<uint32_t*>(lds)[i] = in[i & 511];
    __syncthreads();
    const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    uint32_t la[8];
    for (int ks = 0; ks < 8; ++ks) la[ks] = ldsbase + n * 256 + (((ks * 2 + kg) ^ (n & 15)) << 4);   // the kernel's swizzled fragment address
    const uint32_t m0base = __builtin_amdgcn_readfirstlane(ldsbase + (threadIdx.x >> 6) * 1024);
    const char* g = kglob + (size_t)blockIdx.x * ((size_t)400 * 3 * 32768 + 98304);
    const uint32_t voff = threadIdx.x * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
        : CLOB%(aclob)s, "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "m0", "scc", "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r;
    asm volatile("v_add_f32 %%0, v193, v192" : "=v"(r));
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (lane == 0) cyc[2048 + blockIdx.x * 8 + (threadIdx.x >> 6)] = r1 - r0;   // 100 MHz real-time ticks of the same interval
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
    L.append(f"v_mov_b32 v{UB_M_}, 0xff800000")
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
    for kind in ("afix", "bfix", "abfix", "bpair", "bquad", "apair", "aquad"):
        variants["mfma_" + kind] = dict(reuse=kind)
    for n_, o_ in (("w1_mfma_only", dict(softmax=False)), ("w1_valu_only", dict(mfma=False)), ("w1_math", dict()), ("w1_math_lds", dict(lds=True)),
                   ("w1_math_lds_bar", dict(lds=True, bar=True)), ("w1_math_lds_bar_dma", dict(lds=True, bar=True, dma=True))):
        variants[n_] = dict(o_, w1=True)
    out = [UB_HEAD % dict(clob=clobbers())]
    for name, o in variants.items():
        body = []
        if o.get("w1"):
            for s_ in range(12):
                tile, sub = divmod(s_, 4)
                nxt_tile, nxt_sub = divmod((s_ + 1) % 12, 4)
                dma = None
                if o.get("dma"):   # two 1 KiB pieces per wave and stage (4 waves move the stage's 8 KiB)
                    dst = ((tile + 2) % 3) * 32768 + sub * 8192
                    dma = [f"s_add_u32 m0, s22, {dst}", "s_nop 0", f"global_load_lds_dwordx4 v{DMAV + sub}, s[24:25]",
                           f"s_add_u32 m0, s22, {dst + 4096}", "s_nop 0", f"global_load_lds_dwordx4 v{DMAV + sub}, s[26:27]"]
                    if sub == 3:
                        dma += ["s_add_u32 s24, s24, 0x8000", "s_addc_u32 s25, s25, 0", "s_add_u32 s26, s26, 0x8000", "s_addc_u32 s27, s27, 0"]
                body += w1_stage(s_, o, lds_imm=(nxt_tile, nxt_sub), dma=dma)
                if sub == 3 and o.get("bar"):
                    if o.get("dma"):
                        body.append("s_waitcnt vmcnt(8)")
                    body.append("s_barrier")
            init = [l for l in ub_init() if not any(f"v_mov_b32 v{i}," in l for i in range(64, 128))]
            init += [f"v_accvgpr_write_b32 a{i}, v{32 + (i % 8)}" for i in range(64)] + [f"v_mov_b32 v{i}, 0" for i in range(64, 112)]
            if o.get("dma"):
                init += ["s_mov_b32 s22, %2", "s_mov_b64 s[24:25], %3", "s_add_u32 s26, s24, 0x1000", "s_addc_u32 s27, s25, 0"]
                init += [f"v_add_u32 v{DMAV + i}, {i * 8192}, %1" for i in range(4)]
            out.append(UB_KERNEL % dict(name=name, init=fmt(init), body=fmt(body), iters=400, threads=256,
                                        aclob=", " + ", ".join(f'"a{i}"' for i in range(64))))
            continue
        if o.get("reuse"):
            out.append(UB_KERNEL % dict(name=name, init=fmt(ub_init()), body=fmt(mfma_reuse_body(o["reuse"])), iters=400, threads=512, aclob=""))
            continue
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
        out.append(UB_KERNEL % dict(name=name, init=fmt(init), body=fmt(body), iters=400 // 1, threads=512, aclob=""))
    # host
    out.append(r'''
static const char* g_only = nullptr;   // argv[1]: run only this variant ("all" = every one)
static int g_loop_ms = 0;              // argv[2]: after the one-shot measurement, relaunch the variant back to back for this many ms
                                       // (sustained clocks / power: tools/power_clock_lab.py samples amdsmi beside it)
static int g_mode = -1;                // argv[3]: 0 = 8 workgroups / constant operands only, 1 = 256 workgroups / random operands only
static double now_s() { timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + tv.tv_usec * 1e-6; }
template <typename K> void run(const char* name, K kern, int blocks, const uint32_t* in, const char* kglob, unsigned long long* d_cyc, float* sink, int threads = 512) {
    if (g_only && strcmp(g_only, "all") && strcmp(g_only, name)) return;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(d_cyc, 0, 4096 * 8);
    kern<<<blocks, threads, 98304>>>(d_cyc, sink, in, kglob);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, threads, 98304>>>(d_cyc, sink, in, kglob);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    auto summarize = [&](double& ticks, double& mhz) {
        std::vector<unsigned long long> h(4096);
        hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, c = 0, r = 0; int nz = 0;
        for (int i = 0; i < blocks * 8; ++i) if (h[i]) { a += (double)h[i]; ++nz; if (h[2048 + i]) { c += (double)h[i]; r += (double)h[2048 + i]; } }
        ticks = a / (nz ? nz : 1);
        mhz = r > 0 ? c / r * 100.0 : 0.0;
    };
    const double stages = 400.0 * 12.0;
    double ticks, mhz; summarize(ticks, mhz);
    printf("%-22s blocks %3d: %7.1f ns per stage (%d wave%s/SIMD)   %7.1f s_memtime ticks per stage   kernel %.1f us   in-kernel clock %.0f MHz\n", name, blocks, ms * 1e6 / stages,
           threads / 256, threads == 512 ? "s" : "", ticks / stages, ms * 1e3, mhz);
    if (g_loop_ms > 0) {
        const double t0 = now_s();
        int n = 0;
        hipEventRecord(e0);
        while ((now_s() - t0) * 1e3 < g_loop_ms) { for (int i = 0; i < 8; ++i) kern<<<blocks, threads, 98304>>>(d_cyc, sink, in, kglob); n += 8; hipStreamSynchronize(0); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        const double t1 = now_s();
        hipEventElapsedTime(&ms, e0, e1);
        summarize(ticks, mhz);
        printf("LOOP %s blocks %d t0 %.6f t1 %.6f launches %d ns_per_stage %.1f clock_mhz_last_launch %.0f\n", name, blocks, t0, t1, n, ms * 1e6 / n / stages, mhz);
    }
    fflush(stdout);
}
int main(int argc, char** argv) {
    if (argc > 1) g_only = argv[1];
    if (argc > 2) g_loop_ms = atoi(argv[2]);
    if (argc > 3) g_mode = atoi(argv[3]);
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
        if (g_mode >= 0 && mode != g_mode) continue;
        const int blocks = mode ? 256 : 8;
        const uint32_t* in = mode ? in_rand : in_const;
        printf("== %s\n", mode ? "256 workgroups, random operands" : "8 workgroups, constant operands");
''')
    for name, o in variants.items():
        out.append(f'        run("{name}", k_{name}, blocks, in, kglob, d_cyc, sink, {256 if o.get("w1") else 512});\n')
    out.append("    }\n    return 0;\n}\n")
    return "".join(out)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "ubench"
    if what == "ubench":
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_stage.hip")
        open(path, "w").write(gen_ubench())
        print(path)


# ------------------------------------------------------------------------------------------------------------------
# production loops (kvpress_amd/csrc/snapkv_asm.inc, included by snapkv_mfma.hip)
# ------------------------------------------------------------------------------------------------------------------
# Scalar registers used inside the blocks (all listed as clobbers):
#   s20 c = log2(e)/sqrt(D)        s21 tiles left for the block       s22 LDS address of this wave's 1 KiB slot in ring buffer 0
#   s[24:25] global address of the tile the next request group reads (NBUF - 1 tiles ahead, clamped to the walk's last tile)
#   s26 bytes between a workgroup's tiles      s28 address advances left (clamp)      s29, s30 scratch
#   s[44:45] c in both halves (packed fma)
#   pass 2 only: s27 wave index, s[40:41] global address of the column sums of the tile to flush, s42 its stride, s31 tile barriers passed
import os as _os

SGPR_CLOB = '"s20", "s21", "s22", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s40", "s41", "s42", "s44", "s45", "s46", "vcc", "scc"' + (
    ', "s50", "s51", "s52", "s54", "s55", "s56", "s57", "s58", "s59", "s60"' if int(_os.environ.get("GEN_STAMP", "0")) else "")
GEN_ABL = set(filter(None, _os.environ.get("GEN_ABL", "").split(",")))   # lab builds: nodma, nobar, novalu, nomfma, nolds (results wrong)
# schedule labs (results unchanged): GEN_DMA_AT=k issues a stage's LDS-DMA request after its k-th MFMA instead of at the head;
# GEN_RPS fragment reads per MFMA slot (default 2: slots 0-3), GEN_READ_FROM first slot that carries reads; GEN_PRIO=1 raises
# the priority of waves 4-7 (the younger wave of every SIMD) once, before the loop
GEN_DMA_AT = int(_os.environ["GEN_DMA_AT"]) if _os.environ.get("GEN_DMA_AT", "") != "" else None
GEN_RPS = int(_os.environ.get("GEN_RPS", "2"))
GEN_READ_FROM = int(_os.environ.get("GEN_READ_FROM", "0"))
GEN_PRIO = int(_os.environ.get("GEN_PRIO", "0"))
# GEN_STAMP=1 (with -DKVP_SK_STAMP for snapkv_mfma.hip, tools/build_variants.sh "stamp=..."): pass 1 returns, instead of its statistics,
# the cycles a wave spent in the whole loop / waiting for K fragments (even / odd window rows of part_m) and waiting for the K stream /
# at the tile barrier (even / odd rows of part_z): tools/sk_lab.py --stamps prints the breakdown
GEN_STAMP = int(_os.environ.get("GEN_STAMP", "0"))
# GEN_DMA_NT: non-temporal hint on the K stream's LDS-DMA requests: 1 = pass 1 only, 3 = pass 2 only, 2 = both
GEN_DMA_NT = int(_os.environ.get("GEN_DMA_NT", "0"))


class Cfg:
    """Register map and ring geometry of one pass.  QF / KF0 / KF1 as above; nacc accumulators from v128; then T (16 temporaries),
    pass-specific registers, two sets of 8 LDS fragment addresses (ring buffers 0-1 and 2-3: the ds offset field is 16 bits) and
    the 4 per-lane global offsets of a tile's requests."""

    def __init__(self, nbuf, nacc, extra):
        self.nbuf, self.nacc = nbuf, nacc
        self.acc = [128 + 16 * i for i in range(nacc)]
        v = 128 + 16 * nacc
        self.T = v; v += 16
        self.extra = v; v += extra
        self.laddr = v; v += 8
        self.laddr2 = v; v += 8
        self.dmav = v; v += 4
        self.last = v
        per = 4 * nbuf
        while per % nacc or per % 2:
            per += 4 * nbuf
        self.period = per                      # stages after which ring buffer, accumulator roles and fragment parity repeat
        self.inflight = 4 * (nbuf - 3) + 3     # requests of this wave that may stay in flight at a tile barrier (tiles newer than t + 1)

    def pos(self, p):
        """pattern position p (stage index mod period) -> (ring buffer of its tile, sub-tile)"""
        return (p // 4) % self.nbuf, p % 4


# (a ring of four tiles + four accumulators for pass 1 -- GEN_P1_NBUF=4 -- measured the same as three: 0.2957 vs 0.2969 ms per layer)
# GEN_P1_LAZY (default 1): pass 1 WITHOUT a running maximum per sub-tile ("lazy offset", p1l_* below): four accumulators, ring of three
GEN_P1_LAZY = int(_os.environ.get("GEN_P1_LAZY", "1"))   # production since round 3 (0: the running-maximum loop of round 2)
P1 = Cfg(nbuf=int(_os.environ.get("GEN_P1_NBUF", "3")), nacc=4 if (int(_os.environ.get("GEN_P1_NBUF", "3")) == 4 or GEN_P1_LAZY) else 3, extra=14)
P2 = Cfg(nbuf=3, nacc=3, extra=16 + 2 + 5)
# pass 1 extras: running max m, running sum z, offsets / rescale factors of the X and M parts, partial sums, scratch
# (S0, S1 and the two copies of each offset are even-aligned register pairs for the packed variants)
S0, S1, OFFX, OFFX2, OFFM, OFFM2, M_, Z_, RX, RM, TMAX, MNEW = (P1.extra + (P1.extra & 1) + i for i in range(12))
GEN_PK = set(filter(None, _os.environ.get("GEN_PK", "").split(",")))   # packed f32 variants of the exp part: "fma", "add"
# pass 2 extras: 16 row normalisers, 2 partial sums, LDS write address, flush read address / store offset / value / scratch
AR = P2.extra
P2_S0, P2_S1, REDW, FLR, FLO, FLV, FLT = (P2.extra + 16 + i for i in range(7))
assert AR % 2 == 0 and P2_S0 % 2 == 0 and P1.T % 2 == 0 and P2.T % 2 == 0
RED_BYTES = 3 * 16 * 128 * 4   # three tiles in flight x (8 waves x 2 lane halves) x 128 keys


def stage_head(cfg, p, flush=None):
    """Head of a stage: wait for this stage's K fragments (and LDS writes); at a tile's last sub-tile also for the NEXT tile's
    LDS-DMA (only the requests of newer tiles may still be in flight) + the workgroup barrier that publishes it and retires the
    previous tile's buffer; then the request of sub-block `sub` of the tile NBUF - 1 ahead, which lands in the buffer of the
    previous tile (address clamped to the walk's last tile: s28 counts the advances left)."""
    buf, sub = cfg.pos(p)
    if GEN_STAMP:
        # lab: where a wave's cycles go.  s_memtime before / after each wait of the head; s50 += fragment (lgkmcnt) wait, s51 += K stream
        # (vmcnt) wait, s52 += barrier wait.  Every stamp is an SMEM round trip itself (~50-100 cycles, included in the sums).
        L = ["s_memtime s[56:57]", "s_waitcnt lgkmcnt(0)", "s_memtime s[58:59]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s60, s58, s56", "s_add_u32 s50, s50, s60"]
        if sub == 3:
            L += [f"s_waitcnt vmcnt({cfg.inflight})", "s_memtime s[56:57]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s60, s56, s58", "s_add_u32 s51, s51, s60",
                  "s_barrier", "s_memtime s[58:59]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s60, s58, s56", "s_add_u32 s52, s52, s60"]
            if flush:
                L += flush
    else:
        L = ["s_waitcnt lgkmcnt(0)"]
        if sub == 3:
            L += [f"s_waitcnt vmcnt({cfg.inflight})"] + ([] if "nobar" in GEN_ABL else ["s_barrier"])
            if flush:
                L += flush
    dst = ((buf + cfg.nbuf - 1) % cfg.nbuf) * 32768 + sub * 8192
    D = []
    if "nodma" not in GEN_ABL:
        nt = " nt" if (GEN_DMA_NT == 2 or (GEN_DMA_NT == 1 and cfg is P1) or (GEN_DMA_NT == 3 and cfg is P2)) else ""
        D += [f"s_add_u32 m0, s22, {dst}", "s_nop 0", f"global_load_lds_dwordx4 v{cfg.dmav + sub}, s[24:25]{nt}"]
    if sub == 3:
        D += ["s_cmp_lg_u32 s28, 0", "s_cselect_b32 s29, s26, 0", "s_cselect_b32 s30, 1, 0", "s_sub_u32 s28, s28, s30",
              "s_add_u32 s24, s24, s29", "s_addc_u32 s25, s25, 0"]
    if GEN_DMA_AT is None:
        return L + D, []
    return L, D     # lab: the request is issued after MFMA GEN_DMA_AT of the stage instead of at its head


def prefetch_reads(cfg, p, kfl):
    nbuf, nsub = cfg.pos((p + 1) % cfg.period)
    base = cfg.laddr2 if nbuf >= 2 else cfg.laddr
    imm = nsub * 8192 + (nbuf % 2) * 32768
    return [f"ds_read_b128 {vr(kfl + 4 * ks, 4)}, v{base + ks} offset:{imm}" for ks in range(8)]


def interleave(L, mf, reads, slots, dma=()):
    L, dma = (L[0], L[1]) if isinstance(L, tuple) else (L, list(dma))
    if "novalu" in GEN_ABL:
        slots = [[] for _ in slots]
    if "nomfma" in GEN_ABL:
        mf = ["s_nop 0"] * len(mf)
    if "nolds" in GEN_ABL:
        reads = []
    ri = 0
    for k in range(8):
        L.append(mf[k])
        if dma and k == GEN_DMA_AT:
            L += dma
        for _ in range(GEN_RPS):
            if ri < len(reads) and k >= GEN_READ_FROM:
                L.append(reads[ri])
                ri += 1
        L += slots[k]
    L += reads[ri:]
    return L


# ---- pass 1 -----------------------------------------------------------------------------------------------------------
def exp_sum_ops(T, acc, s0, off1, off2, cs, neg=False):
    """16 x (fma, exp) into T and the two interleaved partial sums s0 / s0+1 (element i goes to sum i & 1).  off1(i): scalar
    offset operand of element i, off2: the register PAIR holding the offsets of elements (2i, 2i+1) -- or a callable -- for the
    packed fma (GEN_PK=fma: half the fma issue slots; GEN_PK=add: v_pk_add_f32 on the sum pair)."""
    n = "-" if neg else ""
    def F(i):   # elements 2i, 2i+1
        if "fma" in GEN_PK:
            o2 = off2(i) if callable(off2) else off2
            mod = " neg_lo:[0,0,1] neg_hi:[0,0,1]" if neg else ""
            return [f"v_pk_fma_f32 v[{T + 2 * i}:{T + 2 * i + 1}], v[{acc + 2 * i}:{acc + 2 * i + 1}], s[44:45], {o2}{mod}"]
        return [f"v_fma_f32 v{T + 2 * i}, {cs}, v{acc + 2 * i}, {n}{off1(2 * i)}", f"v_fma_f32 v{T + 2 * i + 1}, {cs}, v{acc + 2 * i + 1}, {n}{off1(2 * i + 1)}"]
    E = lambda i: [f"v_exp_f32 v{T + 2 * i}, v{T + 2 * i}", f"v_exp_f32 v{T + 2 * i + 1}, v{T + 2 * i + 1}"]
    def A(i):   # pair i
        if i == 0:
            return []                                      # folded into the add of pair 1
        a = f"v[{T}:{T + 1}]" if i == 1 else f"v[{s0}:{s0 + 1}]"
        if "add" in GEN_PK:
            return [f"v_pk_add_f32 v[{s0}:{s0 + 1}], {a}, v[{T + 2 * i}:{T + 2 * i + 1}]"]
        a0, a1 = (f"v{T}", f"v{T + 1}") if i == 1 else (f"v{s0}", f"v{s0 + 1}")
        return [f"v_add_f32 v{s0}, {a0}, v{T + 2 * i}", f"v_add_f32 v{s0 + 1}, {a1}, v{T + 2 * i + 1}"]
    ops = []
    for b in range(4):
        ops += F(2 * b) + F(2 * b + 1) + E(2 * b) + E(2 * b + 1)
        if b >= 1:
            ops += A(2 * (b - 1)) + A(2 * (b - 1) + 1)
    ops += A(6) + A(7)
    return ops


# Offsets / rescale factors / running maxima live in register PAIRS indexed by the parity of the sub-tile they belong to (the
# pattern period is even): the max part of sub-tile j writes OFF[j & 1], R[j & 1], MX[j & 1] and reads MX[(j - 1) & 1]; the exp
# part of sub-tile j reads OFF[j & 1], R[j & 1] one stage later.  No register-to-register moves.
OFF = (OFFX, OFFM)
R = (RX, RM)
MX = (M_, MNEW)


def p1_valu_x(accx, j, cs="s20"):
    """exp / sum part for sub-tile j (accumulator accx): z <- z * 2^(c m_{j-1} - c m_j) + sum_k 2^(c l_k - c m_j)"""
    T = P1.T
    off, r = OFF[j & 1], R[j & 1]
    ops = exp_sum_ops(T, accx, S0, lambda i: f"v{off}", f"v[{off}:{off + 1}]", cs)
    ops.append(f"v_add_f32 v{S0}, v{S0}, v{S1}")
    ops.append(f"v_fma_f32 v{Z_}, v{Z_}, v{r}, v{S0}")
    return ops


def p1_valu_m(accm, j, cs="s20"):
    """row maximum of sub-tile j (accumulator accm) -> running max MX[j & 1], offset OFF[j & 1] = -c m_j, rescale factor R[j & 1]"""
    mprev, mnew = MX[(j - 1) & 1], MX[j & 1]
    ops = [f"v_max3_f32 v{TMAX}, v{accm}, v{accm + 1}, v{accm + 2}"]
    for i in range(3, 15, 2):
        ops.append(f"v_max3_f32 v{TMAX}, v{TMAX}, v{accm + i}, v{accm + i + 1}")
    ops.append(f"v_max3_f32 v{mnew}, v{mprev}, v{TMAX}, v{accm + 15}")
    ops.append(f"v_mul_f32_e64 v{OFF[j & 1]}, {cs}, -v{mnew}")
    if "fma" in GEN_PK:
        ops.append(f"v_mov_b32 v{OFF[j & 1] + 1}, v{OFF[j & 1]}")   # the packed fma wants the offset in both halves of a pair
    ops.append(f"v_fma_f32 v{R[j & 1]}, {cs}, v{mprev}, v{OFF[j & 1]}")
    ops.append(f"v_exp_f32 v{R[j & 1]}, v{R[j & 1]}")
    return ops


def p1_prod_stage(p, do_m=True, do_x=True, dt="bf16"):
    """One pass-1 stage at pattern position p: head, MFMA chain of sub-tile s, and between the MFMAs the fragment reads of s+1,
    the exp / sum of s-2 and the row maximum of s-1 (sub-tile parities = pattern-position parities: the period is even)."""
    n = P1.nacc
    accw, accm, accx = P1.acc[p % n], P1.acc[(p - 1) % n], P1.acc[(p - 2) % n]
    kfu, kfl = (KF0, KF1) if p % 2 == 0 else (KF1, KF0)
    L = stage_head(P1, p)
    x = p1_valu_x(accx, p - 2) if do_x else []
    m = p1_valu_m(accm, p - 1) if do_m else []
    slots = [[] for _ in range(8)]
    if do_x and do_m:
        spread(slots, x[:len(x) // 2] + m[:4] + x[len(x) // 2:] + m[4:])
    elif do_m:
        spread(slots, m, 3, 8)   # the accumulator being reduced finished at the end of the previous stage: keep clear of it
    mf = [mfma(accw, kfu + 4 * k, QF + 4 * k, k == 0, dt) for k in range(8)]
    return interleave(L, mf, prefetch_reads(P1, p, kfl), slots)


def p1_drain(p_end):
    """after the stage at pattern position p_end: exp/sum of the last two sub-tiles, maximum of the last; the final running
    maximum is left in M_ whatever the parity"""
    a_last, a_prev = P1.acc[p_end % P1.nacc], P1.acc[(p_end - 1) % P1.nacc]
    L = p1_valu_x(a_prev, p_end - 1) + p1_valu_m(a_last, p_end) + p1_valu_x(a_last, p_end)
    if MX[p_end & 1] != M_:
        L.append(f"v_mov_b32 v{M_}, v{MX[p_end & 1]}")
    return L


def tile_loop(cfg, stage_fn, drain_fn, first_two):
    """[stage 0][stage 1] LOOP{ stages 2 .. period+1 with an exit check after every tile } + one drain per exit.  s21 = tiles to do (>= 1)."""
    L = list(first_two)
    L.append("1:")
    exits = []
    for p in range(2, cfg.period + 2):
        L += stage_fn(p % cfg.period)
        if p % 4 == 3:
            L += ["s_sub_u32 s21, s21, 1", "s_cmp_eq_u32 s21, 0", f"s_cbranch_scc1 {100 + p}f"]
            exits.append(p)
    L.append("s_branch 1b")
    for p in exits:
        L.append(f"{100 + p}:")
        L += drain_fn(p % cfg.period)
        if p != exits[-1]:
            L.append("s_branch 9f")
    L.append("9:")
    return L


def prio_lines():
    """GEN_PRIO: waves 4-7 (s27 = wave index) get priority 1 for the whole loop; reset at the end of the block"""
    if not GEN_PRIO:
        return []
    return ["s_cmp_lt_u32 s27, 4", "s_cbranch_scc1 48f", "s_setprio 1", "48:"]


# ---- pass 1, lazy offset --------------------------------------------------------------------------------------------------
# The running maximum costs 11 of pass 1's 59 VALU instructions per stage, and the passes are bound by exactly that issue stream.
# Here the exp part uses an offset OFF = -c * m_ref that is only RAISED when a sub-tile's sum overflows a threshold:
#     stage p:   MFMA chain of sub-tile p  ||  exp / sum of sub-tile p-2 with the current offset -> S0 (+ one v_cmp: S0 >= 2^64 or NaN)
#     head of stage p+1:  branch on that compare; fast path z += S0 (1 instruction).
#     slow path (out of line, per pattern position): row maximum of sub-tile p-2 -> m_new = max(m_ref, tmax), z *= 2^(c m_ref - c m_new),
#                redo the sub-tile's exp / sum at the new offset, z += it.  Its accumulator (one of FOUR) is still intact: the chain
#                of stage p+2 is the first to overwrite it.
# m_ref starts at -inf (OFF = +inf): the first sub-tile overflows by construction and initialises it.  Sums stay below 2^64 * 16 per
# sub-tile and every term is >= 2^-126 only if within ~190 log2 units below m_ref -- terms further down are below float32
# resolution of the sum anyway (as with a true running maximum).  The statistics (m_ref * c, z) are a valid (reference, sum) pair
# for softmax_merge whatever m_ref is.  49 VALU per stage instead of 59.
MREF, OFFL, RRL, OFFN = M_, OFFX, RX, OFFM     # registers of the lazy variant (the running-max variant's, renamed)


def p1l_exp_ops(accx):
    ops = exp_sum_ops(P1.T, accx, S0, lambda i: f"v{OFFL}", f"v[{OFFL}:{OFFL + 1}]", "s20")
    ops.append(f"v_add_f32 v{S0}, v{S0}, v{S1}")
    return ops


def p1l_raise(acc):
    """row maximum of `acc` -> m_ref = max(m_ref, tmax), OFF, RRL = 2^(c m_old - c m_new); then the exp / sum at the new offset and
    z = z * RRL + sum (the safe update: valid for every lane)"""
    ops = [f"v_max3_f32 v{TMAX}, v{acc}, v{acc + 1}, v{acc + 2}"]
    for i in range(3, 15, 2):
        ops.append(f"v_max3_f32 v{TMAX}, v{TMAX}, v{acc + i}, v{acc + i + 1}")
    ops.append(f"v_max3_f32 v{MNEW}, v{MREF}, v{TMAX}, v{acc + 15}")
    ops.append(f"v_mul_f32_e64 v{OFFN}, s20, -v{MNEW}")
    ops.append(f"v_fma_f32 v{RRL}, s20, v{MREF}, v{OFFN}")
    ops.append(f"v_exp_f32 v{RRL}, v{RRL}")
    ops.append(f"v_mov_b32 v{MREF}, v{MNEW}")
    ops.append(f"v_mov_b32 v{OFFL}, v{OFFN}")
    ops += p1l_exp_ops(acc)
    ops.append(f"v_fma_f32 v{Z_}, v{Z_}, v{RRL}, v{S0}")
    return ops


def p1l_check(tag, acc_checked, slow_blocks):
    """head-of-stage lines: take the slow path if the pending sub-tile's sum overflowed, else z += S0"""
    slow_blocks.append([f"{200 + tag}:"] + p1l_raise(acc_checked) + [f"s_branch {300 + tag}b"])
    return [f"s_cbranch_vccnz {200 + tag}f", f"v_add_f32 v{Z_}, v{Z_}, v{S0}", f"{300 + tag}:"]


def p1l_stage(p, do_x, do_check, slow_blocks, dt):
    n = P1.nacc
    accw, accx, accc = P1.acc[p % n], P1.acc[(p - 2) % n], P1.acc[(p - 3) % n]
    kfu, kfl = (KF0, KF1) if p % 2 == 0 else (KF1, KF0)
    L, D = stage_head(P1, p)
    if do_check:
        L = L + p1l_check(p, accc, slow_blocks)
    slots = [[] for _ in range(8)]
    if do_x:
        spread(slots, p1l_exp_ops(accx) + [f"v_cmp_nlt_f32_e64 vcc, v{S0}, s46"])
    mf = [mfma(accw, kfu + 4 * k, QF + 4 * k, k == 0, dt) for k in range(8)]
    return interleave((L, D), mf, prefetch_reads(P1, p, kfl), slots)


def p1l_body(dt):
    c = P1
    L = []
    for ks in range(8):
        L.append(f"global_load_dwordx4 {vr(QF + 4 * ks, 4)}, %2, off offset:{ks * 32}")
    for ks in range(8):
        L.append(f"v_mov_b32 v{c.laddr + ks}, %{7 + ks}")
        L.append(f"v_add_u32 v{c.laddr2 + ks}, 0x10000, %{7 + ks}")
    for i in range(4):
        L.append(f"v_mov_b32 v{c.dmav + i}, %{3 + i}")
    L += ["s_mov_b32 s22, %15", "s_mov_b64 s[24:25], %16", "s_mov_b32 s26, %17", "s_mov_b32 s21, %18", "s_mov_b32 s20, %19", "s_mov_b32 s44, %19", "s_mov_b32 s45, %19", "s_mov_b32 s28, %20",
          "s_mov_b32 s46, 0x5f800000",                                       # 2^64: the overflow threshold of a sub-tile's sum
          f"v_mov_b32 v{MREF}, 0xff800000", f"v_mov_b32 v{OFFL}, 0x7f800000", f"v_mov_b32 v{Z_}, 0",   # m_ref = -inf, OFF = +inf
          "s_mov_b32 s27, %21"] + prio_lines() + [
          "s_waitcnt vmcnt(0)", "s_barrier"]
    if GEN_STAMP:
        L += ["s_mov_b32 s50, 0", "s_mov_b32 s51, 0", "s_mov_b32 s52, 0", "s_memtime s[54:55]", "s_waitcnt lgkmcnt(0)"]
    L += [f"ds_read_b128 {vr(KF0 + 4 * ks, 4)}, v{c.laddr + ks}" for ks in range(8)]
    slow = []
    # stages 0, 1: chains only; stage 2: first exp part (no check pending yet); from stage 3 on: check + exp part
    L += p1l_stage(0, False, False, slow, dt) + p1l_stage(1, False, False, slow, dt)
    L.append("1:")
    exits = []
    first = True
    for p in range(2, c.period + 2):
        # the very first pass through position 2 has nothing to check, but its S0 / vcc are undefined: make the check harmless by
        # defining them before the loop (S0 = 0, vcc = 0) -- see below
        L += p1l_stage(p % c.period, True, True, slow, dt)
        if p % 4 == 3:
            L += ["s_sub_u32 s21, s21, 1", "s_cmp_eq_u32 s21, 0", f"s_cbranch_scc1 {100 + p}f"]
            exits.append(p)
    L.append("s_branch 1b")
    for p in exits:
        q = p % c.period
        L.append(f"{100 + p}:")
        # drain after the stage at position q: pending check of sub-tile q-2, then the safe update for sub-tiles q-1 and q
        dslow = []
        L += p1l_check(400 + q - 200, P1.acc[(q - 2) % c.nacc], dslow)      # labels 400+q / 500+q
        L += p1l_raise(P1.acc[(q - 1) % c.nacc]) + p1l_raise(P1.acc[q % c.nacc])
        L.append("s_branch 9f")
        for b in dslow:
            L += b
    for b in slow:
        L += b
    L.append("9:")
    if GEN_STAMP:
        T = P1.T
        L += ["s_memtime s[56:57]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s60, s56, s54",
              f"v_mbcnt_lo_u32_b32 v{T}, -1, 0", f"v_and_b32 v{T}, 1, v{T}", f"v_cmp_eq_u32 vcc, 1, v{T}",
              f"v_mov_b32 v{T + 1}, s60", f"v_mov_b32 v{T + 2}, s50", f"v_cndmask_b32 v{T + 1}, v{T + 1}, v{T + 2}, vcc", f"v_cvt_f32_u32 %0, v{T + 1}",
              f"v_mov_b32 v{T + 1}, s51", f"v_mov_b32 v{T + 2}, s52", f"v_cndmask_b32 v{T + 1}, v{T + 1}, v{T + 2}, vcc", f"v_cvt_f32_u32 %1, v{T + 1}"]
    else:
        L += (["s_setprio 0"] if GEN_PRIO else []) + [f"v_mov_b32 %0, v{MREF}", f"v_mov_b32 %1, v{Z_}"]
    # the first check (position 2 of the first pass) must find S0 = 0 and vcc = 0
    i = L.index("1:")
    L[i:i] = [f"v_mov_b32 v{S0}, 0", "s_mov_b64 vcc, 0"]
    return L


def p1_prod_body(dt):
    if GEN_P1_LAZY:
        return p1l_body(dt)
    return p1_prod_body_runmax(dt)


def p1_prod_body_runmax(dt):
    c = P1
    L = []
    for ks in range(8):
        L.append(f"global_load_dwordx4 {vr(QF + 4 * ks, 4)}, %2, off offset:{ks * 32}")
    for ks in range(8):
        L.append(f"v_mov_b32 v{c.laddr + ks}, %{7 + ks}")
        L.append(f"v_add_u32 v{c.laddr2 + ks}, 0x10000, %{7 + ks}")
    for i in range(4):
        L.append(f"v_mov_b32 v{c.dmav + i}, %{3 + i}")
    L += ["s_mov_b32 s22, %15", "s_mov_b64 s[24:25], %16", "s_mov_b32 s26, %17", "s_mov_b32 s21, %18", "s_mov_b32 s20, %19", "s_mov_b32 s44, %19", "s_mov_b32 s45, %19", "s_mov_b32 s28, %20",
          f"v_mov_b32 v{MX[1]}, 0xff800000", f"v_mov_b32 v{Z_}, 0",      # the max part of sub-tile 0 reads MX[1]
          "s_mov_b32 s27, %21"] + prio_lines() + [
          "s_waitcnt vmcnt(0)", "s_barrier"]
    L += [f"ds_read_b128 {vr(KF0 + 4 * ks, 4)}, v{c.laddr + ks}" for ks in range(8)]
    first = p1_prod_stage(0, do_m=False, do_x=False, dt=dt) + p1_prod_stage(1, do_m=True, do_x=False, dt=dt)
    L += tile_loop(c, lambda p: p1_prod_stage(p, dt=dt), p1_drain, first)
    L += (["s_setprio 0"] if GEN_PRIO else []) + [f"v_mov_b32 %0, v{M_}", f"v_mov_b32 %1, v{Z_}"]
    return L


# ---- pass 2 -----------------------------------------------------------------------------------------------------------
def p2_valu_x(accx, red_imm, cs="s20"):
    """column sums of P = 2^(c * logit - a_row) over this lane's 16 rows for its key -> LDS slot (wave, lane half)"""
    ops = exp_sum_ops(P2.T, accx, P2_S0, lambda i: f"v{AR + i}", lambda i: f"v[{AR + 2 * i}:{AR + 2 * i + 1}]", cs, neg=True)
    ops.append(f"v_add_f32 v{P2_S0}, v{P2_S0}, v{P2_S1}")
    ops.append(f"ds_write_b32 v{REDW}, v{P2_S0} offset:{red_imm}")
    return ops


def p2_red_imm(p_of_subtile):
    """LDS offset (relative to the lane's write address) of the column-sum slot of the sub-tile computed at pattern position p"""
    buf, sub = P2.pos(p_of_subtile % P2.period)
    return buf * 8192 + sub * 128


_label = [50]


def new_label():
    _label[0] += 2
    return _label[0]


def p2_flush(buf):
    """after a tile barrier: waves 0 and 1 add the 16 partial column sums of a finished tile (red buffer `buf`) in a fixed order
    and store its 128 column sums; skipped at the first barrier (s31 = barriers passed: nothing finished yet)."""
    label = new_label()
    L = ["s_cmp_eq_u32 s31, 0", f"s_cbranch_scc1 {label}f", "s_cmp_gt_u32 s27, 1", f"s_cbranch_scc1 {label + 1}f"]
    regs = [FLV, FLT] + [P2.T + i for i in range(14)]      # the T registers are free at a stage head
    for slot in range(16):
        L.append(f"ds_read_b32 v{regs[slot]}, v{FLR} offset:{buf * 8192 + slot * 512}")
    L.append("s_waitcnt lgkmcnt(0)")
    for slot in range(1, 16):
        L.append(f"v_add_f32 v{FLV}, v{FLV}, v{regs[slot]}")
    L += [f"global_store_dword v{FLO}, v{FLV}, s[40:41]", f"{label + 1}:", "s_add_u32 s40, s40, s42", "s_addc_u32 s41, s41, 0", f"{label}:", "s_add_u32 s31, s31, 1"]
    return L


def p2_prod_stage(p, do_x=True, dt="bf16"):
    accw, accx = P2.acc[p % 3], P2.acc[(p - 2) % 3]
    kfu, kfl = (KF0, KF1) if p % 2 == 0 else (KF1, KF0)
    buf, sub = P2.pos(p)
    # at the tile barrier: flush the tile BEFORE this one (its last column sums were written two stages ago)
    L = stage_head(P2, p, p2_flush((buf + 2) % 3) if sub == 3 else None)
    slots = [[] for _ in range(8)]
    if do_x:
        spread(slots, p2_valu_x(accx, p2_red_imm(p - 2)))
    mf = [mfma(accw, QF + 4 * k, kfu + 4 * k, k == 0, dt) for k in range(8)]   # C = Q . K^T: a lane owns one key
    return interleave(L, mf, prefetch_reads(P2, p, kfl), slots)


def p2_drain(p_end):
    a_last, a_prev = P2.acc[p_end % 3], P2.acc[(p_end - 1) % 3]
    buf, _ = P2.pos(p_end)
    L = p2_valu_x(a_prev, p2_red_imm(p_end - 1)) + p2_valu_x(a_last, p2_red_imm(p_end))
    # the last tile's column sums: publish, then flush
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier", "s_mov_b32 s31, 1"]
    L += p2_flush(buf)
    return L


def p2_prod_body(dt):
    c = P2
    L = []
    for ks in range(8):
        L.append(f"global_load_dwordx4 {vr(QF + 4 * ks, 4)}, %0, off offset:{ks * 32}")
    for j in range(4):   # normalisers of rows (r & 3) + 8 (r >> 2) + 4 kg: four runs of four consecutive rows
        L.append(f"global_load_dwordx4 {vr(AR + 4 * j, 4)}, %21, off offset:{j * 32}")
    for ks in range(8):
        L.append(f"v_mov_b32 v{c.laddr + ks}, %{5 + ks}")
        L.append(f"v_add_u32 v{c.laddr2 + ks}, 0x10000, %{5 + ks}")
    for i in range(4):
        L.append(f"v_mov_b32 v{c.dmav + i}, %{1 + i}")
    L += ["s_mov_b32 s22, %13", "s_mov_b64 s[24:25], %14", "s_mov_b32 s26, %15", "s_mov_b32 s21, %16", "s_mov_b32 s20, %17", "s_mov_b32 s44, %17", "s_mov_b32 s45, %17", "s_mov_b32 s28, %18",
          "s_mov_b32 s27, %19", "s_mov_b64 s[40:41], %20", "s_mov_b32 s42, %22", "s_mov_b32 s31, 0",
          f"v_mov_b32 v{REDW}, %23", f"v_mov_b32 v{FLR}, %24", f"v_mov_b32 v{FLO}, %25"] + prio_lines() + [
          "s_waitcnt vmcnt(0)", "s_barrier"]
    L += [f"ds_read_b128 {vr(KF0 + 4 * ks, 4)}, v{c.laddr + ks}" for ks in range(8)]
    first = p2_prod_stage(0, do_x=False, dt=dt) + p2_prod_stage(1, do_x=False, dt=dt)
    L += tile_loop(c, lambda p: p2_prod_stage(p, dt=dt), p2_drain, first)
    return L + (["s_setprio 0"] if GEN_PRIO else [])


INC_HEAD = """// GENERATED by tools/gen_stage_asm.py kernel -- do not edit (tests/test_capi_symbols.py checks it is up to date).
// Hand-scheduled steady-state loops of the SnapKV window-attention passes: see tools/gen_stage_asm.py for the register maps
// and the schedule.
#pragma once
#define KVP_P1_ASM_CLOBBERS %(clob1)s, %(sclob)s, "memory"
#define KVP_P2_ASM_CLOBBERS %(clob2)s, %(sclob)s, "memory"
#define KVP_P2_RED_BYTES %(red)d
#define KVP_P1_NBUF %(nbuf1)d
"""


def gen_kernel_inc():
    out = [INC_HEAD % dict(clob1=clobbers(32, P1.last), clob2=clobbers(32, P2.last), sclob=SGPR_CLOB, red=RED_BYTES, nbuf1=P1.nbuf)]
    for dt in ("bf16", "f16"):
        _label[0] = 50
        for name, body in (("P1", p1_prod_body(dt)), ("P2", p2_prod_body(dt))):
            out.append(f"#define KVP_{name}_ASM_{dt.upper()} \\\n" + " \\\n".join(f'    "{l}\\n"' for l in body) + "\n")
    return "\n".join(out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "kernel":
    path = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "kvpress_amd", "csrc", "snapkv_asm.inc")
    open(path, "w").write(gen_kernel_inc())
    print(path)
