// mhx_device.h -- shared between the HIP kernels (mhx_kernels.hip) and the host
// engine (mhx_engine.cpp).  Internal; the public surface is include/mhx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mhx_device_consts.h"

namespace mhx {

struct HashArgs {
    const uint8_t *base;   // 16-byte aligned; tiles are laid over base + [0, ...)
    uint64_t begin, end;   // logical span, byte offsets from base
    uint32_t tile0;        // first tile of this launch
    uint32_t ntiles;       // tiles in this launch (== gridDim.x)
    uint32_t first_tile;   // first tile of the push (its line phase is 0)
    uint32_t hash32;       // 1: keep the low 32 bits of the hash (k <= 16)
    uint32_t *ticket;      // zeroed before every launch
    uint64_t *tile_state;  // FASTQ look-back words, zeroed before every push
    const uint64_t *thresh; // admission threshold T (inclusive)
    uint64_t *keys;
    uint32_t *cnts;
    uint64_t slot_mask;
    uint64_t *stats;
};

struct TableArgs {
    uint64_t *keys;
    uint32_t *cnts;
    uint64_t nslots;
    uint64_t *thresh;
    uint32_t *hist;      // kHistBins counters, zero between rounds
    uint64_t *acc;       // 2 accumulators of the tighten pass (occupied, solid), zero between rounds
    uint64_t *stats;
    uint32_t min_mult;
    uint32_t sketch_size;
};

// launchers (mhx_kernels.hip)
hipError_t launch_hash(int k, int fmt, const HashArgs &a, hipStream_t st);
hipError_t launch_tighten(const TableArgs &a, hipStream_t st);
hipError_t launch_extract(const TableArgs &a, uint64_t limit, uint32_t min_count, uint64_t *out_keys,
                          uint32_t *out_cnts, uint32_t cap, uint32_t *out_n, hipStream_t st);
bool hash_k_supported(int k);

struct DistArgs {
    const uint64_t *q;
    const uint32_t *q_len;
    const uint64_t *r;
    const uint32_t *r_len;
    uint32_t nq, nr, stride, s;
    int k;
    uint32_t *common, *denom;
    double *dist;
};
hipError_t launch_dist_pairs(const DistArgs &a, hipStream_t st);

} // namespace mhx
