// tools/mfma_shape_probe.hip -- (GPU box) what the int8 matrix pipe SUSTAINS chip-wide with operands from LDS, by tile shape:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_probe tools/mfma_shape_probe.hip && /tmp/mfma_shape_probe
// Variants (one workgroup of four waves per CU, one wave per SIMD, like the multiplier waves of the commit kernels; 256 workgroups):
//   r16   : v_mfma_i32_16x16x64_i8, 13 x 3 tiles, operands in registers (no LDS traffic): the pipe alone
//   l16   : the same MFMAs, A (13 reads) and B (3 reads) of 16 bytes per lane from LDS per K-step -- the digit-plane commit kernel's blocking
//   l32   : v_mfma_i32_32x32x32_i8, 3 x 3 tiles (144 accumulator registers), two K-halves per K-step: 2 x (3 + 3) reads for 18 MFMAs
//   l32b  : 32x32x32, 5 x 2 tiles (160 accumulator registers): 2 x (5 + 2) reads for 20 MFMAs
// Reported: int8 TOP/s (2 ops per MAC) over the launch, shader clock (cycles of workgroup 0 over the 100 MHz real-time counter), LDS operand bytes per kMAC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ unsigned long long g_clk[4][2];

template <int MODE>
__global__ void __launch_bounds__(256) k_probe(int iters, int *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned i = threadIdx.x; i < 48 * 1024 / 4; i += 256) ((unsigned *)smem)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned char *base = smem + lane * 16 + wave * 1024;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int sink = 0;
    if (MODE == 0 || MODE == 1) {
        v4i acc[13][3];
        for (int m = 0; m < 13; m++) for (int n = 0; n < 3; n++) acc[m][n] = v4i{0, 0, 0, 0};
        v4i ar = *(const v4i *)base, br = *(const v4i *)(base + 4096);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int s = 0; s < 3; s++) {
                v4i b[3];
#pragma unroll
                for (int n = 0; n < 3; n++) b[n] = MODE == 1 ? *(const v4i *)(base + 32768 + ((it + s * 3 + n) & 7) * 1024) : br;
#pragma unroll
                for (int m = 0; m < 13; m++) {
                    const v4i a = MODE == 1 ? *(const v4i *)(base + ((it * 3 + s * 13 + m) & 31) * 1024) : ar;
#pragma unroll
                    for (int n = 0; n < 3; n++) acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[n], acc[m][n], 0, 0, 0);
                }
            }
        }
        for (int m = 0; m < 13; m++) for (int n = 0; n < 3; n++) sink += acc[m][n].x + acc[m][n].w;
    } else if (MODE == 2) {
        v16i acc[3][3];
        for (int m = 0; m < 3; m++) for (int n = 0; n < 3; n++) for (int q = 0; q < 16; q++) acc[m][n][q] = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int s = 0; s < 6; s++) {      // 3 K-steps of 64 = 6 halves of 32
                v4i a[3], b[3];
#pragma unroll
                for (int n = 0; n < 3; n++) b[n] = *(const v4i *)(base + 32768 + ((it + s * 3 + n) & 7) * 1024);
#pragma unroll
                for (int m = 0; m < 3; m++) a[m] = *(const v4i *)(base + ((it * 3 + s * 3 + m) & 31) * 1024);
#pragma unroll
                for (int m = 0; m < 3; m++)
#pragma unroll
                    for (int n = 0; n < 3; n++) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], acc[m][n], 0, 0, 0);
            }
        }
        for (int m = 0; m < 3; m++) for (int n = 0; n < 3; n++) sink += acc[m][n][0] + acc[m][n][15];
    } else {
        v16i acc[5][2];
        for (int m = 0; m < 5; m++) for (int n = 0; n < 2; n++) for (int q = 0; q < 16; q++) acc[m][n][q] = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int s = 0; s < 6; s++) {
                v4i a[5], b[2];
#pragma unroll
                for (int n = 0; n < 2; n++) b[n] = *(const v4i *)(base + 32768 + ((it + s * 2 + n) & 7) * 1024);
#pragma unroll
                for (int m = 0; m < 5; m++) a[m] = *(const v4i *)(base + ((it * 3 + s * 5 + m) & 31) * 1024);
#pragma unroll
                for (int m = 0; m < 5; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], acc[m][n], 0, 0, 0);
            }
        }
        for (int m = 0; m < 5; m++) for (int n = 0; n < 2; n++) sink += acc[m][n][0] + acc[m][n][15];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[MODE][0] = __builtin_amdgcn_s_memtime() - c0; g_clk[MODE][1] = __builtin_amdgcn_s_memrealtime() - r0; }
    if (sink == 0x7fffffff) out[threadIdx.x] = sink;
}

template <int MODE>
static int run(const char *name, double macs_per_iter_wave, double lds_bytes_per_iter_wave, int iters, int *out) {
    (void)hipFuncSetAttribute((const void *)k_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);   // 150 KB: one workgroup per CU
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_probe<MODE>, dim3(256), dim3(256), 150 * 1024, 0, iters, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long clk[4][2];
        CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
        const double macs = macs_per_iter_wave * iters * 4 * 256;
        if (rep) printf("%-5s %8.3f ms  %7.1f int8 TOP/s  shader clock %5.0f MHz  %5.1f cycles per 16 kMAC and SIMD  LDS operand bytes per kMAC %.1f\n", name, ms, 2 * macs / ms / 1e9,
                        (double)clk[MODE][0] / ((double)clk[MODE][1] / 100.0), (double)clk[MODE][0] / (macs_per_iter_wave * iters / 16384.0), lds_bytes_per_iter_wave / (macs_per_iter_wave / 1000.0));
    }
    return 0;
}

int main() {
    int *out;
    CK(hipMalloc(&out, 4096));
    const int iters = 20000;
    if (run<0>("r16", 3 * 39 * 16384.0, 0.0, iters, out)) return 1;
    if (run<1>("l16", 3 * 39 * 16384.0, 3 * 16 * 1024.0, iters, out)) return 1;
    if (run<2>("l32", 6 * 9 * 32768.0, 6 * 6 * 1024.0, iters, out)) return 1;
    if (run<3>("l32b", 6 * 10 * 32768.0, 6 * 7 * 1024.0, iters, out)) return 1;
    return 0;
}
