// tools/fetch_rate.hip -- does the size of a VALU loop body (instruction fetch) or the dependence between its
// instructions change the issue rate?  One workgroup of 256*w threads per CU (w waves per SIMD), a loop whose body is
// BODY instructions of the hash loop's classes (v_mul_lo_u32, v_alignbit_b32, v_add3_u32, v_mad_u64_u32-free mix),
// the same total instruction count for every BODY; CHAIN = 1: every instruction depends on the previous one (one
// chain per lane), CHAIN = 0: eight independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define I_MUL(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##n) : "s"(k1));
#define I_ALN(n) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a##n) : "v"(b));
#define I_AD3(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "s"(k2));
#define I_XOR(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define G8I(n0, n1, n2, n3, n4, n5, n6, n7) I_MUL(n0) I_ALN(n1) I_AD3(n2) I_XOR(n3) I_MUL(n4) I_ALN(n5) I_AD3(n6) I_ALN(n7)
#define G8_IND G8I(0, 1, 2, 3, 4, 5, 6, 7)
#define G8_DEP G8I(0, 0, 0, 0, 0, 0, 0, 0)
#define R2(X) X X
#define R4(X) R2(X) R2(X)
#define R8(X) R4(X) R4(X)
#define R16(X) R8(X) R8(X)
#define R32(X) R16(X) R16(X)
#define R64(X) R32(X) R32(X)
#define R128(X) R64(X) R64(X)
#define R256(X) R128(X) R128(X)

constexpr int kTotal = 1 << 17; // instructions per wave

template <int BODY8, int CHAIN> __global__ void k(uint64_t *out, uint32_t seed, uint32_t k1, uint32_t k2)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed * 2654435761u + threadIdx.x;
    const int iters = kTotal / (BODY8 * 8);
    for (int i = 0; i < iters; ++i) {
        if (CHAIN) {
            if (BODY8 == 2) { R2(G8_DEP) } else if (BODY8 == 32) { R32(G8_DEP) } else if (BODY8 == 128) { R128(G8_DEP) } else { R256(G8_DEP) R256(G8_DEP) }
        } else {
            if (BODY8 == 2) { R2(G8_IND) } else if (BODY8 == 32) { R32(G8_IND) } else if (BODY8 == 128) { R128(G8_IND) } else { R256(G8_IND) R256(G8_IND) }
        }
    }
    const uint32_t sink = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (sink == 0x12345678u) out[0] = sink;
}

template <int BODY8, int CHAIN> void run(uint64_t *d, double ghz)
{
    printf("body %5d instr (%6d B)  %s:", BODY8 * 8, BODY8 * 8 * 8, CHAIN ? "one chain " : "8 chains  ");
    for (int w = 1; w <= 8; w += (w < 4 ? 1 : 2)) {
        const int blocks = 256 * (w > 4 ? 2 : 1), threads = 256 * (w > 4 ? w / 2 : w); // w > 4: two workgroups per CU
        hipLaunchKernelGGL((k<BODY8, CHAIN>), dim3(blocks), dim3(threads), 0, 0, d, 77u, 0x9E3779B1u, 0x85EBCA6Bu);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<BODY8, CHAIN>), dim3(blocks), dim3(threads), 0, 0, d, 78u, 0x9E3779B1u, 0x85EBCA6Bu);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: w waves x kTotal instructions in ms
        printf("  w%d %5.2f", w, (ms - 0.006) * 1e-3 * ghz * 1e9 / ((double)w * kTotal));
    }
    printf("   cycles/instr/SIMD at the given clock\n");
}

int main(int argc, char **argv)
{
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    uint64_t *d; hipMalloc(&d, 64);
    run<2, 0>(d, ghz);   run<2, 1>(d, ghz);
    run<32, 0>(d, ghz);  run<32, 1>(d, ghz);
    run<128, 0>(d, ghz); run<128, 1>(d, ghz);
    run<512, 0>(d, ghz); run<512, 1>(d, ghz);
    return 0;
}
