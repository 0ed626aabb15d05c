#!/usr/bin/env python3
"""Lab build with in-kernel clock stamps: kvpress_amd/lib/variants/clocklab.so.

The production sources carry no measurement code.  This script PATCHES copies of snapkv_mfma.hip and gather.hip (text
inserted after anchor lines that must exist exactly once) so that the first lane of every workgroup of snapkv_p1_asm,
snapkv_p2_asm and gather_vec_kernel stores (s_memtime, s_memrealtime) at its start and at its end plus the XCC id, and exports
readers for the stamp tables.  Effective shader clock of a workgroup = (shader-cycle ticks) / (100 MHz real-time ticks) * 100 MHz
(the rule of kvp_clock_probe).  tools/power_clock_lab.py uses the variant; nothing in the product loads it.

    python tools/make_clock_lab.py          # needs kvpress_amd/build/*.o of the production build (python -m kvpress_amd.build)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kvpress_amd", "csrc")
OUT = os.path.join(ROOT, "kvpress_amd", "lib", "variants", "clocklab.so")
HIPCC = "/opt/rocm/bin/hipcc"

STAMP_DECL = r'''
// ---- lab (tools/make_clock_lab.py): per-workgroup clock stamps ----
#define KVP_LAB_MAXWG 16384
__device__ unsigned long long kvp_lab_tab[%(nk)d][KVP_LAB_MAXWG][4];
__device__ unsigned int kvp_lab_xcc[%(nk)d][KVP_LAB_MAXWG];
__device__ __forceinline__ uint32_t kvp_lab_wg() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
__device__ __forceinline__ void kvp_lab_begin(int kind) {
    if (threadIdx.x == 0 && kvp_lab_wg() < KVP_LAB_MAXWG) {
        unsigned int xcc;
        asm volatile("s_getreg_b32 %%0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        kvp_lab_xcc[kind][kvp_lab_wg()] = xcc & 0xf;
        kvp_lab_tab[kind][kvp_lab_wg()][0] = __builtin_amdgcn_s_memtime();
        kvp_lab_tab[kind][kvp_lab_wg()][1] = __builtin_amdgcn_s_memrealtime();
    }
}
__device__ __forceinline__ void kvp_lab_end(int kind) {
    if (threadIdx.x == 0 && kvp_lab_wg() < KVP_LAB_MAXWG) {
        kvp_lab_tab[kind][kvp_lab_wg()][2] = __builtin_amdgcn_s_memtime();
        kvp_lab_tab[kind][kvp_lab_wg()][3] = __builtin_amdgcn_s_memrealtime();
    }
}
'''

READER = r'''
// lab reader: copies kind's table ([KVP_LAB_MAXWG][4] u64) and xcc ids ([KVP_LAB_MAXWG] u32) to the host and clears them
extern "C" int %(name)s(int kind, unsigned long long* tab_host, unsigned int* xcc_host) {
    if (kind < 0 || kind >= %(nk)d) return -1;
    if (hipMemcpyFromSymbol(tab_host, HIP_SYMBOL(kvp_lab_tab), sizeof(unsigned long long) * KVP_LAB_MAXWG * 4, sizeof(unsigned long long) * KVP_LAB_MAXWG * 4 * kind) != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(xcc_host, HIP_SYMBOL(kvp_lab_xcc), sizeof(unsigned int) * KVP_LAB_MAXWG, sizeof(unsigned int) * KVP_LAB_MAXWG * kind) != hipSuccess) return -3;
    static unsigned long long zeros[KVP_LAB_MAXWG * 4];
    if (hipMemcpyToSymbol(HIP_SYMBOL(kvp_lab_tab), zeros, sizeof(zeros), sizeof(zeros) * kind) != hipSuccess) return -4;
    return 0;
}
'''


def insert_after(src, anchor, text, nth=0, count=None):
    """insert `text` after the nth occurrence of the line containing `anchor` (count: required number of occurrences)"""
    lines = src.split("\n")
    hits = [i for i, l in enumerate(lines) if anchor in l]
    if count is not None and len(hits) != count:
        raise SystemExit(f"make_clock_lab: anchor {anchor!r} found {len(hits)} times, expected {count}")
    if not hits:
        raise SystemExit(f"make_clock_lab: anchor {anchor!r} not found")
    i = hits[nth]
    lines.insert(i + 1, text)
    return "\n".join(lines)


def insert_before(src, anchor, text, nth=0, count=None):
    lines = src.split("\n")
    hits = [i for i, l in enumerate(lines) if anchor in l]
    if count is not None and len(hits) != count:
        raise SystemExit(f"make_clock_lab: anchor {anchor!r} found {len(hits)} times, expected {count}")
    i = hits[nth]
    lines.insert(i, text)
    return "\n".join(lines)


def patch_snapkv(src):
    src = insert_after(src, "typedef float f32x16", STAMP_DECL % dict(nk=2), count=1)
    # pass 1 (asm): begin after the ring declaration, end before the final merge of the two lane halves
    src = insert_after(src, "__shared__ __attribute__((aligned(16))) unsigned char lds[NB * MF_TILEB];", "    kvp_lab_begin(0);", count=1)
    src = insert_before(src, "float mm = m == KVP_NEG_INF ? KVP_NEG_INF : m * c, zz = z;", "    kvp_lab_end(0);", nth=1, count=2)
    # pass 2 (asm): begin after its reduction-slot declaration; end = the kernel's last statement
    src = insert_after(src, "__shared__ __attribute__((aligned(16))) unsigned char red3[KVP_P2_RED_BYTES];", "    kvp_lab_begin(1);", count=1)
    marker = "namespace {\n__global__ __launch_bounds__(256) void add_slab_kernel"
    if src.count(marker) != 1:
        raise SystemExit("make_clock_lab: add_slab_kernel marker")
    head, tail = src.split(marker)
    k = head.rindex("    wait_all_landed();\n}")
    head = head[:k] + "    wait_all_landed();\n    kvp_lab_end(1);\n}" + head[k + len("    wait_all_landed();\n}"):]
    src = head + marker + tail
    return src + READER % dict(name="kvp_lab_stamps_snapkv", nk=2)


def patch_gather(src):
    src = insert_after(src, "constexpr int GA_UNROLL = 4;", STAMP_DECL % dict(nk=1), count=1)
    src = insert_after(src, "constexpr int GPB = GA_THREADS / LPR;", "    kvp_lab_begin(0);", count=1)
    # end of gather_vec_kernel: the closing of its row loop is followed by the scalar kernel's comment
    marker = "// Any row size / alignment: element-granular copy"
    if src.count(marker) != 1:
        raise SystemExit("make_clock_lab: gather marker")
    head, tail = src.split(marker)
    k = head.rindex("}\n}")
    head = head[:k] + "}\n    kvp_lab_end(0);\n}" + head[k + 3:]
    src = head + marker + tail
    return src + READER % dict(name="kvp_lab_stamps_gather", nk=1)


def main():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs = []
    for name, fn, extra in (("snapkv_mfma", patch_snapkv, ["-mllvm", "-amdgpu-mfma-vgpr-form"]), ("gather", patch_gather, [])):
        src = fn(open(os.path.join(CSRC, name + ".hip")).read())
        tmp = f"/tmp/clocklab_{name}.hip"
        open(tmp, "w").write(src)
        obj = f"/tmp/clocklab_{name}.o"
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), *extra, "-c", tmp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        objs.append(obj)
    bdir = os.path.join(ROOT, "kvpress_amd", "build")
    rest = [os.path.join(bdir, f) for f in sorted(os.listdir(bdir)) if f.endswith(".o") and not f.startswith("contrib_") and f not in ("snapkv_mfma.o", "gather.o")]
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *rest, *objs], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    print(OUT)


if __name__ == "__main__":
    main()
