// tools/cndmask_rate.hip -- microbenchmark: cost of lane-select forms (v_cndmask with VCC / SGPR-pair
// mask vs a bit-select through v_bitop3_b32 / v_bfi_b32 with the mask in a VGPR) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 tools/cndmask_rate.hip -o /tmp/cndmask_rate
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
    uint32_t m = (threadIdx.x & 1) ? 0xFFFFFFFFu : 0u;
    uint64_t sm;
    asm volatile("v_cmp_ne_u32 vcc, 0, %1\n s_mov_b64 %0, vcc" : "=s"(sm) : "v"(m) : "vcc");
    uint64_t t0 = clock64();
    for (int i = 0; i < ITERS; ++i) {
#define A(n) a##n
        if (OP == 0) {
#define X(n) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 1) {
#define X(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "s"(sm));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 2) {
#define X(n) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca" : "+v"(A(n)) : "v"(b), "v"(m));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 3) {
#define X(n) asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(A(n)) : "v"(b), "v"(m));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 4) {
#define X(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 5) {
#define X(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 6) {
#define X(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 7) {
#define X(n) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 8) {
#define X(n) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 9) {
#define X(n) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(A(n)) : "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if (OP == 10) {
#define X(n) asm volatile("v_mov_b32 %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 11) {
#define X(n) asm volatile("v_or_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 12) {
#define X(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A(n)) : "s"(0x15151515));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 13) {
#define X(n) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(A(n)));
            REP8(X) REP8(X)
#undef X
        }
    }
    uint64_t t1 = clock64();
    uint64_t sink = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (sink == 0x123456789ull) out[1u << 20] = sink;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP> void run(const char *name, uint64_t *d_out)
{
    printf("%-28s", name);
    for (int wps = 1; wps <= 4; ++wps) {
        const int threads = 256 * wps, blocks = 256;
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 77u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 78u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int nw = blocks * threads / 64;
        std::vector<uint64_t> h(nw);
        hipMemcpy(h.data(), d_out, nw * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += (double)v;
        const double per_wave = sum / nw / (ITERS * 16.0);
        printf("  w%d: %6.2f (%5.2f/SIMD) %6.3fms", wps, per_wave, per_wave / wps, ms);
    }
    printf("\n");
}
int main()
{
    uint64_t *d_out; hipMalloc(&d_out, (1u << 20) * 8 + 64);
    run<4>("v_xor_b32", d_out);
    run<5>("v_and_b32", d_out);
    run<11>("v_or_b32", d_out);
    run<12>("v_xor_b32 sgpr-const", d_out);
    run<10>("v_mov_b32", d_out);
    run<13>("v_lshrrev_b32 imm", d_out);
    run<0>("v_cndmask_e32 vcc", d_out);
    run<1>("v_cndmask_e64 sgpr pair", d_out);
    run<2>("v_bitop3 select (vgpr mask)", d_out);
    run<3>("v_bfi_b32", d_out);
    run<6>("v_and_or_b32", d_out);
    run<7>("v_alignbyte_b32 imm", d_out);
    run<8>("v_alignbit_b32 vgpr shift", d_out);
    run<9>("v_cmp + v_addc", d_out);
    return 0;
}
