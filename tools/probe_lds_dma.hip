// Probe: where does global_load_lds_dwordx4 put each lane's 16 bytes?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_lds_dma.hip -o tools/probe_lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t* g, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    // lane L of wave w fetches the 16 B chunk number 1000*w + (63 - L)  (reversed, to tell lane order from address order)
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t* src = g + (size_t)(1000 * w + (63 - lane)) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + w * 2048 + 64), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}
int main() {
    std::vector<uint32_t> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = i;  // chunk c holds words 4c..4c+3
    uint32_t *g, *o; (void)hipMalloc(&g, 65536); (void)hipMalloc(&o, 16384);
    (void)hipMemcpy(g, h.data(), 65536, hipMemcpyHostToDevice);
    k<<<1, 128, 16384>>>(g, o);
    std::vector<uint32_t> r(4096);
    (void)hipMemcpy(r.data(), o, 16384, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
        printf("wave %d, LDS words from its base+64B (expect lane L's chunk = %d + 63 - L at byte 64 + 16 L):\n", w, 1000 * w);
        for (int L = 0; L < 66; ++L) {
            const uint32_t* p = &r[(w * 2048 + 64) / 4 + L * 4];
            if (L < 4 || L > 61) printf("  slot %2d: chunk %u (words %u %u %u %u)\n", L, p[0] / 4, p[0], p[1], p[2], p[3]);
        }
        printf("  word before base+64: %08x\n", r[(w * 2048 + 64) / 4 - 1]);
    }
    return 0;
}
