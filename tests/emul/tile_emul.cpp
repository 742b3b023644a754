// tests/emul/tile_emul.cpp -- CPU phase emulator of sketch_tile_kernel (test tool).
// Runs the host+device functions of auriclass_amd/csrc/mhx_tile.h thread by thread, phase
// by phase, exactly in the order the kernel separates them with __syncthreads(), so that
// the tile logic (masks, look-back arithmetic, work list, canonical k-mer, Murmur) can be
// checked against the oracle without a GPU.  Not part of the product; built by
// tests/test_tile_emulation.py with g++.
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../auriclass_amd/csrc/mhx_tile.h"

using namespace mhx;

struct VecInserter {
    std::vector<uint64_t> *out;
    void operator()(uint64_t h) { out->push_back(h); }
};

template <int K, bool FASTQ>
static void run_tiles(const uint8_t *base, uint64_t begin, uint64_t end, uint64_t T, bool hash32,
                      std::vector<uint64_t> &out, uint64_t *stats)
{
    static TileSmem sm;
    static ThreadState st[kBlock];
    const uint32_t first_tile = (uint32_t)(begin / kTileBytes);
    const uint32_t ntiles = (uint32_t)((end + kTileBytes - 1) / kTileBytes);
    uint32_t line_prefix = 0;
    uint8_t prev_rec = 0;
    VecInserter ins{&out};
    for (uint32_t tile = first_tile; tile < ntiles; ++tile) {
        const uint64_t tile_off = (uint64_t)tile * kTileBytes;
        for (int t = 0; t < kBlock; ++t) phase_stage(sm, t, base, tile_off, end);
        const bool interior = tile_off >= begin && tile_off + kTileBytes + kHaloBytes <= end;
        for (int t = 0; t < kBlock; ++t) phase_classify(sm, t, st[t], tile_off, begin, end, interior);
        uint32_t excl[kBlock] = {0}, tile_total = 0;
        if (FASTQ) { // the kernel's block_scan_excl
            for (int t = 0; t < kBlock; ++t) { excl[t] = tile_total; tile_total += st[t].nlcount; }
        }
        const uint32_t line_base = line_prefix; // what the look-back returns
        line_prefix += tile_total;
        bool bad = false;
        const uint64_t span_left = end > tile_off ? end - tile_off : 0;
        const uint32_t check_limit = span_left < (uint64_t)(kTileBytes + kHaloBytes) ? (uint32_t)span_left : (uint32_t)(kTileBytes + kHaloBytes);
        if (FASTQ) { // the self-synchronising form of the kernel: the phase a tile finds by itself, and the chain check
            const uint32_t self = tile == first_tile ? 0u : phase_selfsync(sm, check_limit);
            if (self == 4u) ++stats[6];                          // tiles left to the look-back pass
            else if (self != (line_base & 3u)) stats[5] |= 1u;   // a phase that is not the running line count's
            // after the repair pass every tile has a record; the tiles that needed it carry the look-back's phase
            const uint8_t rec = phase_record(self == 4u ? line_base : self, tile_total);
            if (tile != first_tile && phase_chain_broken(prev_rec, rec)) stats[5] |= 2u; // what phase_verify_kernel flags
            prev_rec = rec;
        }
        for (int t = 0; t < kBlock; ++t) stats[kStatRecords] += phase_good<FASTQ>(sm, t, st[t], line_base, excl[t], tile_total, check_limit, bad, tile_off, end, (uint32_t)K);
        if (bad) stats[kStatFlags] |= kFlagBadFastq;
        uint32_t ex2[kBlock], run = 0;
        for (int t = 0; t < kBlock; ++t) { uint32_t items = 0; stats[kStatKmers] += phase_runs<K>(sm, t, items); ex2[t] = run; run += items; }
        for (int t = 0; t < kBlock; ++t) phase_compact(sm, t, ex2[t]);
        const uint32_t nitems = sm.misc[1];
        // the candidate queue behind the work list, as on the device; odd tiles get a queue of 5 entries so that the
        // "queue full: finish at once" path runs too
        struct Queue {
            TileSmem &sm; uint32_t first, cap; uint64_t T; VecInserter &ins; uint64_t *inserts;
            void operator()(uint32_t group, int window) const
            {
                const uint32_t code = (group << 3) | (uint32_t)window;
                const uint32_t slot = sm.misc[7]++;
                if (slot < cap) sm.list[first + slot] = (uint16_t)code;
                else *inserts += process_deferred<K>(sm, code, T, ins);
            }
        };
        sm.misc[7] = 0;
        uint32_t qcap = (uint32_t)kGroupsPerTile - nitems;
        if ((tile & 1u) && qcap > 5u) qcap = 5u;
        Queue queue{sm, nitems, qcap, T, ins, &stats[kStatInserts]};
        if (tile % 3u == 2u) { // every third tile: the form that finishes its candidates where they are found
            for (uint32_t it = 0; it < nitems; ++it) stats[kStatInserts] += process_group<K, false>(sm, sm.list[it], T, admission_limit(T), ins, queue);
        } else {
            for (uint32_t it = 0; it < nitems; ++it) process_group<K, true>(sm, sm.list[it], T, admission_limit(T), ins, queue);
            const uint32_t ncand = sm.misc[7] < qcap ? sm.misc[7] : qcap;
            for (uint32_t c = 0; c < ncand; ++c) stats[kStatInserts] += process_deferred<K>(sm, sm.list[nitems + c], T, ins);
        }
        stats[kStatLines] += tile_total;
    }
}

#define K_LIST(X) X(1) X(2) X(3) X(4) X(5) X(7) X(8) X(9) X(11) X(15) X(16) X(17) X(20) X(21) X(24) X(27) X(31) X(32)

extern "C" int emul_sketch(const uint8_t *base, uint64_t begin, uint64_t end, int k, int fmt, uint64_t T,
                           uint64_t *out, uint64_t cap, uint64_t *n_out, uint64_t *stats8)
{
    std::vector<uint64_t> v;
    memset(stats8, 0, 8 * sizeof(uint64_t));
    const bool hash32 = k <= 16;
    switch (k) {
#define X(KK) case KK: if (fmt == 1) run_tiles<KK, true>(base, begin, end, T, hash32, v, stats8); else run_tiles<KK, false>(base, begin, end, T, hash32, v, stats8); break;
        K_LIST(X)
#undef X
    default: return -1;
    }
    *n_out = v.size();
    if (v.size() > cap) return -2;
    if (!v.empty()) memcpy(out, v.data(), v.size() * 8);
    return 0;
}
