// diagnostic: where do the waves of co-resident workgroups land?  (SIMD balance of 7- vs 6-wave blocks with 62 KB of LDS each)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
__global__ void k(unsigned *out, int spin) {
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the block resident for a while so that co-residency is real
    unsigned long long t0 = clock64();
    volatile unsigned char *p = smem;
    while (clock64() - t0 < (unsigned long long)spin) p[threadIdx.x] = (unsigned char)hw;
    if ((threadIdx.x & 63) == 0) {
        unsigned w = threadIdx.x / 64, nw = blockDim.x / 64;
        out[((size_t)blockIdx.x * nw + w) * 2] = hw;
        out[((size_t)blockIdx.x * nw + w) * 2 + 1] = xcc;
    }
}
int main(int argc, char **argv) {
    int nthr = argc > 1 ? atoi(argv[1]) : 448, blocks = 1024, lds = 63632;
    unsigned *d;
    size_t nw = nthr / 64;
    hipMalloc(&d, blocks * nw * 8);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(nthr), lds, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * nw * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per (xcc, se, cu): count waves per SIMD over the first wave of co-resident blocks (blocks < 512)
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < 512; b++)
        for (size_t w = 0; w < nw; w++) {
            unsigned hw = h[(b * nw + w) * 2], xcc = h[(b * nw + w) * 2 + 1] & 15;
            unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 15, se = (hw >> 13) & 7;
            unsigned key = (xcc << 8) | (se << 4) | cuid;
            auto &v = cu[key];
            if (v.empty()) v.assign(4, 0);
            v[simd]++;
        }
    std::map<std::string, int> hist;
    for (auto &kv : cu) {
        char buf[64];
        snprintf(buf, sizeof buf, "%d,%d,%d,%d", kv.second[0], kv.second[1], kv.second[2], kv.second[3]);
        hist[buf]++;
    }
    printf("threads/block %d: %zu CUs seen; waves per SIMD pattern -> #CUs\n", nthr, cu.size());
    for (auto &kv : hist) printf("  %s : %d\n", kv.first.c_str(), kv.second);
    // first two blocks' SIMD sequences
    for (int b = 0; b < 3; b++) {
        printf("  block %d:", b);
        for (size_t w = 0; w < nw; w++) { unsigned hw = h[(b * nw + w) * 2]; printf(" cu%u.s%u", (hw >> 8) & 15, (hw >> 4) & 3); }
        printf("\n");
    }
    return 0;
}
