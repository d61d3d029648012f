// Host micro-probe (no GPU): sustained issue rate of 512-bit vpmadd52luq, vfmadd231pd, vpmuludq and their mixes on this CPU (12 independent
// chains each, registers only).  g++ -O2 -mavx512f -mavx512ifma -mavx512dq tools/ifma_probe.cc -o tools/psv/ifma_probe
#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include <x86intrin.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define I(k) "vpmadd52luq %%zmm30, %%zmm31, %%zmm" #k "\n\t"
#define F(k) "vfmadd231pd %%zmm28, %%zmm29, %%zmm" #k "\n\t"
#define M(k) "vpmuludq %%zmm30, %%zmm31, %%zmm" #k "\n\t"
#define A(k) "vpaddq %%zmm30, %%zmm" #k ", %%zmm" #k "\n\t"
#define S(k) "vpsrlq $3, %%zmm" #k ", %%zmm" #k "\n\t"
#define CLOB "zmm0","zmm1","zmm2","zmm3","zmm4","zmm5","zmm6","zmm7","zmm8","zmm9","zmm10","zmm11","zmm12","zmm13","zmm14","zmm15","zmm16","zmm17","zmm18","zmm19","zmm20","zmm21","zmm22","zmm23","zmm24","zmm25","zmm26","zmm27"
#define RUN(name, n_instr, BODY)                                                          \
    {                                                                                     \
        long n = N;                                                                       \
        double t0 = now();                                                                \
        asm volatile("1:\n\t" BODY "dec %0\n\tjnz 1b\n\t" : "+r"(n) : : CLOB, "cc");    \
        double dt = now() - t0;                                                           \
        printf("%-28s %.4f ns per instruction  (%.2f per ns)\n", name, dt / ((double)N * n_instr) * 1e9, (double)N * n_instr / dt * 1e-9); \
    }
int main() {
    const long N = 30000000;
    asm volatile("vpxorq %%zmm30,%%zmm30,%%zmm30\n\tvpxorq %%zmm31,%%zmm31,%%zmm31\n\tvpxorq %%zmm28,%%zmm28,%%zmm28\n\tvpxorq %%zmm29,%%zmm29,%%zmm29\n\t" ::: "zmm28", "zmm29", "zmm30", "zmm31");
    RUN("ifma x12", 12, I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11))
    RUN("fma x12", 12, F(12) F(13) F(14) F(15) F(16) F(17) F(18) F(19) F(20) F(21) F(22) F(23))
    RUN("ifma x12 + fma x12", 24, I(0) F(12) I(1) F(13) I(2) F(14) I(3) F(15) I(4) F(16) I(5) F(17) I(6) F(18) I(7) F(19) I(8) F(20) I(9) F(21) I(10) F(22) I(11) F(23))
    RUN("ifma x8 + fma x16", 24, I(0) F(12) F(13) I(1) F(14) F(15) I(2) F(16) F(17) I(3) F(18) F(19) I(4) F(20) F(21) I(5) F(22) F(23) I(6) F(24) F(25) I(7) F(26) F(27))
    RUN("vpmuludq x12", 12, M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11))
    RUN("ifma x12 + vpmuludq x12", 24, I(0) M(12) I(1) M(13) I(2) M(14) I(3) M(15) I(4) M(16) I(5) M(17) I(6) M(18) I(7) M(19) I(8) M(20) I(9) M(21) I(10) M(22) I(11) M(23))
    RUN("ifma x12 + vpaddq x12", 24, I(0) A(12) I(1) A(13) I(2) A(14) I(3) A(15) I(4) A(16) I(5) A(17) I(6) A(18) I(7) A(19) I(8) A(20) I(9) A(21) I(10) A(22) I(11) A(23))
    RUN("ifma x12 + vpsrlq x12", 24, I(0) S(12) I(1) S(13) I(2) S(14) I(3) S(15) I(4) S(16) I(5) S(17) I(6) S(18) I(7) S(19) I(8) S(20) I(9) S(21) I(10) S(22) I(11) S(23))
    RUN("vpaddq x12", 12, A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11))
    unsigned aux; uint64_t c0 = __rdtscp(&aux); double s0 = now(); while (now() - s0 < 0.05) {} uint64_t c1 = __rdtscp(&aux);
    printf("tsc %.2f GHz\n", (c1 - c0) / 0.05 / 1e9);
    return 0;
}
