// tools/class_rate.hip -- wall-clock issue cost of single VALU instruction classes (cycles per wave64 instruction per
// SIMD at a given clock), eight independent chains per lane, 256-instruction loop body, w waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/class_rate.hip -o tools/class_rate ; run on the GPU box: tools/class_rate [GHz]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define R2(X) X X
#define R4(X) R2(X) R2(X)
#define R8(X) R4(X) R4(X)
#define R32(X) R8(X) R8(X) R8(X) R8(X)
#define EIGHT(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
constexpr int kTotal = 1 << 17;

#define OP0(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define OP1(n) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a##n) : "s"(k1));
#define OP2(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define OP3(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define OP4(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##n) : "s"(k1));
#define OP5(n) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a##n) : "v"(b));
#define OP6(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define OP7(n) asm volatile("v_mad_u64_u32 %0, s[0:1], %1, %2, 0" : "=v"(q##n) : "v"(a##n), "s"(k1)); a##n = (uint32_t)(q##n >> 32);
#define OP8(n) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q##n) : "v"(qb));
#define OP9(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "s"(mask));
#define OP10(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define OP11(n) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a##n) : "v"(b), "v"(c));
#define OP12(n) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a##n));
#define OP13(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define OP14(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define OP15(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define OP16(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define OP17(n) asm volatile("v_lshl_add_u32 %0, %0, 2, %0" : "+v"(a##n));
#define OP18(n) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define OP19(n) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define OP20(n) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define OP21(n) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define OP22(n) asm volatile("v_mov_b32 %0, %1" : "=v"(a##n) : "v"(b));
#define OP23(n) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define OP24(n) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define OP25(n) asm volatile("v_mad_u64_u32 %0, s[0:1], %1, %2, %0" : "+v"(q##n) : "v"(a##n), "v"(b));

template <int OP> __global__ void k(uint64_t *out, uint32_t seed, uint32_t k1, uint64_t mask)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed * 2654435761u + threadIdx.x, c = b ^ 0x5bd1e995u;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7, qb = ((uint64_t)b << 32) | c;
    for (int i = 0; i < kTotal / 256; ++i) {
#define CASE(N) if (OP == N) { R32(EIGHT(OP##N)) }
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
        CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25)
    }
    const uint64_t sink = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;
    if (sink == 0x12345678u) out[0] = sink;
}

template <int OP> void run(const char *name, uint64_t *d, double ghz)
{
    printf("%-34s", name);
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = 256 * (w > 4 ? 2 : 1), threads = 256 * (w > 4 ? w / 2 : w);
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, d, 77u, 0x9E3779B1u, 0x5555aaaa3333ccccull);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, d, 78u, 0x9E3779B1u, 0x5555aaaa3333ccccull);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  w%d %5.2f", w, (ms - 0.006) * 1e-3 * ghz * 1e9 / ((double)w * kTotal));
    }
    printf("\n");
}

int main(int argc, char **argv)
{
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    uint64_t *d; hipMalloc(&d, 64);
    printf("cycles per wave64 instruction per SIMD at %.2f GHz (wall clock), w waves per SIMD\n", ghz);
    run<0>("v_xor_b32 vgpr", d, ghz);            run<1>("v_xor_b32 sgpr operand", d, ghz);
    run<2>("v_add_u32", d, ghz);                 run<22>("v_mov_b32", d, ghz);
    run<12>("v_lshlrev_b32 imm", d, ghz);        run<17>("v_lshl_add_u32", d, ghz);
    run<3>("v_mul_lo_u32 vgpr", d, ghz);         run<4>("v_mul_lo_u32 sgpr operand", d, ghz);
    run<14>("v_mul_hi_u32", d, ghz);             run<15>("v_mul_u32_u24", d, ghz);
    run<16>("v_mad_u32_u24", d, ghz);            run<7>("v_mad_u64_u32 (addend 0)", d, ghz);
    run<25>("v_mad_u64_u32 (64-bit accumulate)", d, ghz);
    run<5>("v_alignbit_b32 imm shift", d, ghz);  run<23>("v_alignbit_b32 vgpr shift", d, ghz);
    run<6>("v_add3_u32", d, ghz);                run<8>("v_lshl_add_u64", d, ghz);
    run<9>("v_cndmask_b32_e64 sgpr mask", d, ghz); run<10>("v_perm_b32", d, ghz);
    run<11>("v_bitop3_b32", d, ghz);             run<13>("v_and_or_b32", d, ghz);
    run<18>("v_xad_u32", d, ghz);                run<19>("v_pk_add_u16", d, ghz);
    run<20>("v_pk_mul_lo_u16", d, ghz);          run<21>("v_pk_mad_u16", d, ghz);
    run<24>("v_dot4_u32_u8", d, ghz);
    return 0;
}
