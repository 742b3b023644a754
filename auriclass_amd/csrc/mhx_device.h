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
    uint32_t *need_lookback; // FMT 2: set to 1 by a tile that could not find its line phase by itself (and did nothing else)
    uint32_t repair;         // FMT 1: repair pass -- tiles that CAN find their phase by themselves only publish it
    uint32_t queue_candidates; // kernel form: windows that pass the admission test are queued and finished after the hash loop (large sketches)
    uint8_t *phase_rec;      // FMT 2 / repair: one phase_record() per tile of the span (0: phase unknown), for phase_verify_kernel
};

struct TableArgs {
    uint64_t *keys;
    uint32_t *cnts;
    uint64_t nslots;
    uint64_t *thresh;
    uint32_t *hist;      // kHistBins counters, zero between rounds
    uint64_t *acc;       // kAccReplicas x 8 words: accumulators of the tighten pass (occupied, solid), zero between rounds
    uint64_t *stats;
    uint32_t *done;      // ticket of the tighten pass (its last workgroup computes the new threshold), zero between rounds
    uint32_t *need_lookback; // the sketcher's "repair pass due" word (HashArgs::need_lookback), reported by the extract kernel
    uint32_t min_mult;
    uint32_t sketch_size;
    uint32_t sample;     // tighten pass looks at one 256-slot block in `sample` (1 = exact pass)
    uint64_t next_cap;   // tighten pass, m > 1: byte-count cap to apply behind the new threshold (0: none), see cap_threshold_kernel
};

// launchers (mhx_kernels.hip)
hipError_t launch_hash(int k, int fmt, const HashArgs &a, hipStream_t st);
hipError_t launch_tighten(const TableArgs &a, hipStream_t st);
hipError_t launch_reset(const TableArgs &a, uint64_t t_init, uint32_t *tickets, uint32_t ntickets, uint32_t *out_n, hipStream_t st);
hipError_t launch_cap_threshold(uint64_t *thresh, uint64_t cap, uint64_t *stats, hipStream_t st);
hipError_t launch_order_block(const uint64_t *blk, uint32_t cap, uint32_t log2_buckets, uint32_t *cursor, uint32_t *starts,
                              uint32_t *group_total, uint64_t *grouped, uint64_t *host_blk, hipStream_t st);
hipError_t launch_phase_verify(const uint8_t *rec, uint32_t ntiles, uint64_t *stats, hipStream_t st);
hipError_t launch_extract(const TableArgs &a, uint64_t limit, uint32_t min_count, uint64_t *out_keys,
                          uint32_t *out_cnts, uint32_t cap, uint32_t *out_n, uint64_t *flags_out, const uint64_t *limit_dev,
                          uint64_t *limit_out, uint64_t *maxkey_out, hipStream_t st, uint32_t *order_cursor = nullptr, uint32_t order_log2 = 0,
                          uint64_t *hdr_dev = nullptr, uint64_t *hdr_host = nullptr, uint32_t *ticket = nullptr,
                          uint64_t *occ_out = nullptr, uint32_t nhdr = 4, uint64_t *hdr_copy = nullptr);

// sharded path: the other ranks' gathered partial results go into this rank's candidate table (slab_insert_kernel)
constexpr uint32_t kMaxMergeRanks = 64; // ranks per launch (more: several launches)
struct SlabMergeArgs {
    const uint64_t *slabs;  // device: nranks slabs of slab_words 8-byte words each: [hdr_words of header] hashes[cap] | counts u32[cap]
    uint64_t slab_words, cap;
    uint32_t hdr_words;
    uint64_t n[kMaxMergeRanks]; // valid entries of each slab (0: skip)
    uint32_t nranks, own_rank;  // own_rank: slab to skip (already in the table); >= nranks: none
    uint64_t t_min;
    uint64_t maxkey_others;     // occurrences of the hash value 2^64-1 on the other ranks
    uint64_t *keys;
    uint32_t *cnts;
    uint64_t slot_mask;
    uint64_t *thresh;
    uint64_t *stats;
};
hipError_t launch_slab_insert(const SlabMergeArgs &a, uint64_t max_n, hipStream_t st);

// sharded path, the usual case: the gathered slabs are binned by value and merged bin by bin in LDS (mhx_merge.hip)
constexpr uint32_t kMergeMaxBins = 16384, kMergeMaxSlots = 4096; // LDS: 2 x 4 bytes per bin in the scatter pass, 12 per slot in the bin pass
struct MergeArgs {
    const uint64_t *slabs;      // device: nranks slabs of slab_words 8-byte words each: [hdr_words of header] hashes[cap] | counts u32[cap]
    uint64_t slab_words, cap;
    uint32_t hdr_words;
    uint64_t n[kMaxMergeRanks]; // valid entries of each slab
    uint32_t nranks;
    uint32_t min_mult;
    uint64_t t_min;
    uint32_t shift;             // bin of a hash = hash >> shift (< nbins for every hash <= t_min)
    uint32_t nbins;             // power of two, 256 .. kMergeMaxBins
    uint32_t region;            // entries a bin's region holds
    uint32_t table_slots;       // LDS table of merge_bin_kernel (power of two, >= 4/3 region)
    uint32_t chunk;             // slab entries per workgroup of the scatter pass (set by launch_merge_bins)
    uint32_t *cursor;           // [nbins] entries placed per bin; zero between merges (merge_bin_kernel clears it)
    uint32_t *qn;               // [nbins] qualifying entries per bin
    uint32_t *flags;            // [0]: 1 a region overflowed, 2 a table overflowed, 4 too many qualifying entries in a bin; zero between merges
    uint64_t *sc_keys;          // [nbins * region]
    uint32_t *sc_cnts;          // [nbins * region]
};
hipError_t launch_merge_bins(const MergeArgs &a, uint64_t max_n, uint64_t *out, uint32_t out_cap, hipStream_t st);
bool hash_k_supported(int k);

// FASTA on the device (mhx_fasta.hip): raw file bytes -> dense sequence stream + record separator positions.
// ws: fasta_workspace_bytes(n) bytes; after the launch *(uint64_t *)(ws + o_off + 8 * ntiles) is the stream size,
// ((uint32_t *)(ws + o_flags))[0] the format flag (1: FASTQ syntax seen), [1] the number of separators.
constexpr int kFastaTile = 16384;
size_t fasta_workspace_bytes(uint64_t n, size_t *o_summary, size_t *o_in, size_t *o_off, size_t *o_flags);
hipError_t launch_fasta_compact(const uint8_t *base, uint64_t n, uint8_t *ws, uint8_t *out, uint64_t *seps, uint32_t seps_cap, hipStream_t st);

struct DistArgs {
    const uint64_t *q;
    const uint32_t *q_len;
    const uint64_t *r;
    const uint32_t *r_len;
    uint32_t nq, nr, stride, s;
    int k;
    uint32_t *common, *denom; // [nq][out_stride], this call fills columns out_off .. out_off + nr - 1
    double *dist;
    uint32_t out_stride, out_off;
};
hipError_t launch_dist_pairs(const DistArgs &a, hipStream_t st);

// all-vs-refs fast path (nr <= 32): value-range partition + LDS hash probe
#ifndef MHX_DIST_RANGES
#define MHX_DIST_RANGES 1024
#endif
#ifndef MHX_DIST_SLOTS
#define MHX_DIST_SLOTS 2048
#endif
constexpr int kDistRanges = MHX_DIST_RANGES;     // value ranges the hash space is cut into
constexpr int kDistTableSlots = MHX_DIST_SLOTS; // LDS table of one range (refs' hashes of that range)
#ifndef MHX_DIST_QCHUNKS
#define MHX_DIST_QCHUNKS 4
#endif
constexpr int kDistQueryChunks = MHX_DIST_QCHUNKS;   // query chunks per range (grid.y of the range kernel)
constexpr int kDistSegs = 16;         // finish kernel: ranges are summed in 16 segments first
struct DistWork {
    uint32_t *offs_q;   // [nq][kDistRanges + 1] first index of every range in each query list
    uint32_t *offs_r;   // [nr][kDistRanges + 1]
    uint8_t *cpart;     // [nq][kDistRanges][4 * ceil(nr / 4)] shared hashes per (query, range, ref), one byte each
    uint32_t *params;   // [0] shift, [1] overflow flag
};
size_t dist_work_bytes(uint32_t nq, uint32_t nr, size_t *off_q, size_t *off_r, size_t *off_c, size_t *off_p);
hipError_t launch_dist_ranges(const DistArgs &a, const DistWork &w, hipStream_t st);

} // namespace mhx
