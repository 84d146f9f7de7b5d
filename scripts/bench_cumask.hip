// Which XCDs do the bits of a hipExtStreamCreateWithCUMask mask select?  For a few masks: a kernel of 2048 small workgroups on the masked
// stream counts its workgroups per XCC_ID (and per CU id within the XCD).
// build: hipcc --offload-arch=gfx950 -O2 scripts/bench_cumask.hip -o scripts/bench_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void where_kernel(unsigned* per_xcc, unsigned* per_cu) {
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        xcc &= 0xf;
        const unsigned cu = (hwid >> 8) & 0xf, sh = (hwid >> 12) & 0x1, se = (hwid >> 13) & 0x7;
        atomicAdd(per_xcc + xcc, 1u);
        atomicAdd(per_cu + xcc * 128 + se * 32 + sh * 16 + cu, 1u);
    }
    // keep the compute unit busy for a moment so that the workgroups spread over everything the mask allows
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}
int main() {
    unsigned *d_xcc, *d_cu;
    CK(hipMalloc(&d_xcc, 16 * sizeof(unsigned)));
    CK(hipMalloc(&d_cu, 16 * 128 * sizeof(unsigned)));
    struct Case { const char* name; uint32_t mask[8]; };
    std::vector<Case> cases;
    auto make = [&](const char* name, auto pred) { Case c; c.name = name; for (int w = 0; w < 8; w++) c.mask[w] = 0; for (int b = 0; b < 256; b++) if (pred(b)) c.mask[b / 32] |= 1u << (b % 32); cases.push_back(c); };
    make("bits 0..31", [](int b) { return b < 32; });
    make("bits 32..63", [](int b) { return b >= 32 && b < 64; });
    make("bits b % 8 == 0", [](int b) { return b % 8 == 0; });
    make("bits b % 8 == 3", [](int b) { return b % 8 == 3; });
    make("bits 0..7", [](int b) { return b < 8; });
    make("all but bits 0..31", [](int b) { return b >= 32; });
    make("all but b % 8 == 0", [](int b) { return b % 8 != 0; });
    for (const Case& c : cases) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, c.mask) != hipSuccess) { printf("%-22s: mask refused\n", c.name); (void)hipGetLastError(); continue; }
        CK(hipMemsetAsync(d_xcc, 0, 16 * sizeof(unsigned), s));
        CK(hipMemsetAsync(d_cu, 0, 16 * 128 * sizeof(unsigned), s));
        hipLaunchKernelGGL(where_kernel, dim3(2048), dim3(64), 0, s, d_xcc, d_cu);
        unsigned hx[16], hc[16 * 128];
        CK(hipMemcpyAsync(hx, d_xcc, sizeof hx, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(hc, d_cu, sizeof hc, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        printf("%-22s: workgroups per XCC", c.name);
        for (int x = 0; x < 8; x++) printf(" %4u", hx[x]);
        printf("   distinct units per XCC");
        for (int x = 0; x < 8; x++) { int n = 0; for (int k = 0; k < 128; k++) n += hc[x * 128 + k] ? 1 : 0; printf(" %2d", n); }
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    return 0;
}
