// tools/hash_rate.hip -- isolates the arithmetic of the sketch kernel from its memory phases:
//   (1) murmur3_h1<21> alone, (2) the whole work item (process_group<21>) on an LDS-resident tile.
// Reports cycles per wave per window at 1..4 workgroups (of 256 threads) per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../auriclass_amd/csrc/mhx_tile.h"
using namespace mhx;

struct CountCand { uint64_t *acc; __device__ void operator()(uint32_t g, int w) const { *acc += g + w; } };
struct CountIns { unsigned long long *sink; __device__ void operator()(uint64_t h) { atomicAdd(sink, (unsigned long long)h); } };

template <int MODE> __global__ __launch_bounds__(256) void k(unsigned long long *out, int iters, uint64_t T, uint32_t desync)
{
    __shared__ TileSmem sm;
    const int tid = threadIdx.x;
    uint32_t *b = reinterpret_cast<uint32_t *>(sm.bytes);
    for (int i = tid; i < (kTileBytes + kHaloBytes) / 4; i += 256) {
        uint32_t x = i * 2654435761u + 12345u, v = 0;
        for (int j = 0; j < 4; ++j) { v |= (uint32_t)("ACGT"[(x >> (j * 7 + 3)) & 3]) << (8 * j); }
        b[i] = v;
    }
    for (int i = tid; i < kGroupsPerTile / 4; i += 256) sm.valid[i] = 0xFFFFFFFFu;
    __syncthreads();
    CountIns ins{out};
    uint64_t acc = 0;
    if (desync) { // break the lockstep of the waves: each starts at its own time
        const uint32_t wv = blockIdx.x * 4 + tid / 64;
        const uint32_t wait = (wv * 2654435761u) % desync;
        const uint64_t w0 = clock64();
        while (clock64() - w0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    const uint64_t t0 = clock64();
    if (MODE == 0) {
        uint32_t w[8] = {b[tid], b[tid + 1], b[tid + 2], b[tid + 3], b[tid + 4], b[tid + 5] & 0xFF, 0, 0};
        for (int it = 0; it < iters * 8; ++it) {
            const uint64_t h = murmur3_h1<21>(w);
            w[0] ^= (uint32_t)h; w[3] += (uint32_t)(h >> 32);
            acc += h;
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            const uint32_t g = (tid + it * 256) % (kGroupsPerTile - 8);
            CountCand cc{&acc}; acc += process_group<21, false>(sm, g, T, admission_limit(T), ins, cc);
        }
    }
    const uint64_t t1 = clock64();
    if (acc == 0x1234567) out[1] = acc;
    if ((tid & 63) == 0) out[2 + blockIdx.x * 4 + tid / 64] = t1 - t0;
}

template <int MODE> void run(const char *name, unsigned long long *d, uint32_t desync)
{
    const int iters = 256;
    printf("%-28s\n", name);
    for (int bpc = 1; bpc <= 7; ++bpc) {
        const int blocks = 256 * bpc;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0ull, desync);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0ull, desync);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 + blocks * 4);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks * 4; ++i) s += (double)h[2 + i];
        const double per_window = s / (blocks * 4) / (iters * 8.0);
        const double windows = (double)blocks * 256.0 * iters * 8.0;
        printf("  %d wg/CU: %7.1f ticks/window/wave (%6.1f per SIMD)  %.3f ms  %.1f G windows/s  %.2f ns per wave-window per SIMD\n", bpc, per_window, per_window / bpc, ms,
               windows / (ms * 1e-3) / 1e9, ms * 1e6 / (windows / 64.0 / 1024.0));
    }
    printf("\n");
}

int main()
{
    unsigned long long *d; hipMalloc(&d, (2 + 2048 * 4) * 8); hipMemset(d, 0, (2 + 2048 * 4) * 8);
    run<0>("murmur3_h1<21> only", d, 0);
    run<1>("process_group<21> (8 windows)", d, 0);
    run<1>("process_group, waves desynchronised", d, 50000);
    return 0;
}
