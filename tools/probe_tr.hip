// Probe: what does ds_read_b64_tr_b16 deliver?  LDS holds u16 value = its own index; every lane supplies the byte address of
// an 8-byte piece; the four 16-bit results per lane are printed as LDS indices.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_tr.hip -o tools/probe_tr && tools/probe_tr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k(uint32_t* out, int mode) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t l = threadIdx.x;
    uint32_t piece;
    if (mode == 0) piece = l;                               // lane l -> piece l (8 bytes each, contiguous)
    else piece = (l & 15) * 32 + (l >> 4) * 1;              // lane (l&15) -> row (l&15) of a 256-byte-pitch image, group g -> piece g in the row
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + piece * 8;
    uint32_t r0, r1;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n s_waitcnt lgkmcnt(0)\n" : "=v"(*(uint64_t*)&r0), "=v"(r1) : "v"(addr));
    out[l * 2] = r0; out[l * 2 + 1] = r1;
}
__global__ void k2(uint32_t* out, int mode) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t l = threadIdx.x;
    uint32_t piece = mode == 0 ? l : (l & 15) * 32 + (l >> 4);
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + piece * 8;
    uint64_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"(addr));
    out[l * 2] = (uint32_t)r; out[l * 2 + 1] = (uint32_t)(r >> 32);
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 8);
    uint32_t h[128];
    for (int mode = 0; mode < 2; ++mode) {
        k2<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (%s)\n", mode, mode == 0 ? "lane l supplies piece l: u16 indices 4l..4l+3" : "lane supplies piece (l&15)*32 + (l>>4): row pitch 128 u16");
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %4u %4u %4u %4u", l, h[2 * l] & 0xFFFF, h[2 * l] >> 16, h[2 * l + 1] & 0xFFFF, h[2 * l + 1] >> 16);
            if (l % 2 == 1) printf("\n");
        }
    }
    return 0;
}
