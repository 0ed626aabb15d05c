// Which CUs does bit i of a hipExtStreamCreateWithCUMask mask enable on an 8-XCD MI355X?  Launches a spinning kernel of many workgroups
// on streams with different masks and prints, per mask, how many distinct CUs of every XCD ran a workgroup (XCC_ID / HW_ID registers).
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_map.hip -o /tmp/cumask_map && /tmp/cumask_map
// Lab tool (VERDICT r5 #2 probe), not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <set>
#include <vector>

__global__ void where_kernel(uint32_t* out, uint32_t spin_ticks) {
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
        const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int blocks = 2048;
    uint32_t* d;
    hipMalloc(&d, blocks * 8);
    hipMemsetAsync(d, 0xff, blocks * 8, s);
    where_kernel<<<blocks, 512, 65536, s>>>(d, 2000);   // 64 KiB of LDS: at most two workgroups per CU; 20 us each
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::set<uint32_t> cus[8];
    for (int i = 0; i < blocks; ++i) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        if (xcc < 8) cus[xcc].insert((hw >> 8) & 0xff);
    }
    int total = 0;
    printf("%-34s CUs used per XCD:", name);
    for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); total += (int)cus[x].size(); }
    printf("   total %d\n", total);
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    auto bits = [](auto pred) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
    run("all 256 bits", bits([](int) { return true; }));
    run("bits 0..31", bits([](int i) { return i < 32; }));
    run("bits 0..127", bits([](int i) { return i < 128; }));
    run("bits 128..255", bits([](int i) { return i >= 128; }));
    run("even bits", bits([](int i) { return i % 2 == 0; }));
    run("bits with i % 8 == 0", bits([](int i) { return i % 8 == 0; }));
    run("bits with i % 8 < 4", bits([](int i) { return i % 8 < 4; }));
    run("bits with (i / 8) % 2 == 0", bits([](int i) { return (i / 8) % 2 == 0; }));
    run("bits with (i / 8) < 8", bits([](int i) { return (i / 8) < 8; }));
    run("bits with (i / 32) % 2 == 0", bits([](int i) { return (i / 32) % 2 == 0; }));
    return 0;
}
