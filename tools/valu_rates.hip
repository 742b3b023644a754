// tools/valu_rates.hip -- microbenchmark: issue cost (cycles per wave64 instruction per SIMD)
// of the integer VALU instructions the sketch kernel is made of, at 1/2/3/4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o tools/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ void rate_kernel(uint64_t *out, uint32_t seed)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed * 2654435761u + 12345u, c = seed ^ 0x9E3779B9u;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;
    uint64_t t0 = clock64();
    for (int i = 0; i < ITERS; ++i) {
#define A(n) a##n
#define Q(n) q##n
        if (OP == 0) {
#define X(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 1) {
#define X(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 2) {
#define X(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 3) {
#define X(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(Q(n)) : "v"(b), "v"(c) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if (OP == 4) {
#define X(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 5) {
#define X(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 6) {
#define X(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 7) {
#define X(n) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 8) {
#define X(n) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(Q(n)));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 9) {
#define X(n) asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(Q(n)));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 10) {
#define X(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 11) {
#define X(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 12) {
#define X(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A(n)) : "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if (OP == 13) {
#define X(n) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 14) {
#define X(n) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A(n)) : "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if (OP == 15) {
#define X(n) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(A(n)));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 16) {
#define X(n) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(A(n)) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 17) {
#define X(n) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 18) {
#define X(n) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(A(n)));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 19) {
#define X(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(A(n)) : "s"(0x114253d5));
            REP8(X) REP8(X)
#undef X
        }
    }
    uint64_t t1 = clock64();
    uint64_t sink = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;
    if (sink == 0x123456789ull) out[1u << 20] = sink;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP> void run(const char *name, uint64_t *d_out)
{
    printf("%-22s", name);
    for (int wps = 1; wps <= 4; ++wps) { // waves per SIMD
        const int threads = 256 * wps, blocks = 256;
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 77u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 78u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int nw = blocks * threads / 64;
        std::vector<uint64_t> h(nw);
        hipMemcpy(h.data(), d_out, nw * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += (double)v;
        const double per_wave = sum / nw / (ITERS * 16.0);      // clock64 ticks per instruction seen by one wave
        printf("  w%d: %6.2f tick/inst/wave (%5.2f per SIMD) %6.3fms", wps, per_wave, per_wave / wps, ms);
    }
    printf("\n");
}

int main()
{
    uint64_t *d_out; hipMalloc(&d_out, (1u << 20) * 8 + 64);
    run<0>("v_add_u32", d_out);
    run<11>("v_xor_b32", d_out);
    run<15>("v_lshlrev_b32", d_out);
    run<10>("v_add3_u32", d_out);
    run<16>("v_bitop3_b32", d_out);
    run<17>("v_lshl_or_b32", d_out);
    run<18>("v_bfe_u32", d_out);
    run<12>("v_cndmask_b32", d_out);
    run<14>("v_cmp+v_cndmask", d_out);
    run<6>("v_perm_b32", d_out);
    run<7>("v_alignbit_b32", d_out);
    run<1>("v_mul_lo_u32", d_out);
    run<19>("v_mul_lo_u32 sgpr", d_out);
    run<2>("v_mul_hi_u32", d_out);
    run<3>("v_mad_u64_u32", d_out);
    run<4>("v_mul_u32_u24", d_out);
    run<13>("v_mul_hi_u32_u24", d_out);
    run<5>("v_mad_u32_u24", d_out);
    run<8>("v_lshl_add_u64", d_out);
    run<9>("v_lshrrev_b64", d_out);
    return 0;
}
