// mhx_device_consts.h -- geometry and flag constants shared by kernels, host engine and
// the CPU phase emulator (no HIP headers needed).
#pragma once
#include <stdint.h>

namespace mhx {

// ---- tile geometry of the sketch kernel ------------------------------------------------
#ifndef MHX_TILE_BYTES
#define MHX_TILE_BYTES 16384
#endif
constexpr int kTileBytes = MHX_TILE_BYTES;           // bytes of the stream one workgroup owns
constexpr int kHaloBytes = 64;                       // staged beyond the tile (>= 8 + 32)
#ifndef MHX_BLOCK
#define MHX_BLOCK 256
#endif
constexpr int kBlock = MHX_BLOCK;                    // threads per workgroup
constexpr int kGroup = 8;                            // k-mer start positions per work item
constexpr int kGroupsPerTile = kTileBytes / kGroup;  // 2048
constexpr int kBytesPerThread = kTileBytes / kBlock; // 128
constexpr int kWordsPerThread = kBytesPerThread / 32; // 32-byte words classified per thread
static_assert(kBytesPerThread % 32 == 0 && kTileBytes % (16 * kBlock) == 0, "tile geometry");

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull; // vacant slot of the candidate table

// device flag bits (stats[kStatFlags])
constexpr uint64_t kFlagTableFull = 1;   // probe limit hit: result not exact, retry bigger
constexpr uint64_t kFlagBadFastq = 2;    // a record violated the 4-line layout
constexpr uint64_t kFlagSpinTimeout = 4; // look-back spin bound hit (should never happen)
constexpr uint64_t kFlagNeedLookback = 8; // not an error: some FASTQ tile could not find its line phase by itself (repair pass due)
// state bits the extract kernel adds to the flags word it reports (same values as MHX_SLAB_* of include/mhx.h)
constexpr uint64_t kFlagStateBounded = 0x100, kFlagStateEstablished = 0x200;
constexpr uint64_t kFlagErrorMask = 0xFF;

enum Stat : int {
    kStatKmers = 0,   // windows hashed: k-mer starts with K bytes inside one sequence line / record (A/C/G/T or not)
    kStatInserts = 1, // occurrences admitted (hash <= threshold)
    kStatLines = 2,   // newline count of the FASTQ stream
    kStatFlags = 3,
    kStatMaxKey = 4,  // occurrences of the hash value 2^64-1 (cannot live in the table)
    kStatOccupied = 5,
    kStatSolid = 6,   // entries <= T with count >= m seen by the last tighten pass
    kStatRecords = 7, // FASTQ records whose sequence line holds >= k bytes (mash's sequence count)
    kStatStamp0 = 8,  // diagnostic builds (-DMHX_STAMPS): cycles per phase, summed over workgroups (8..13)
    kStatEstablished = 14, // replica 0 only, m > 1: a tighten pass has lowered T from solid (count >= m) hashes
    kStatBounded = 15,     // replica 0 only, m > 1: the byte-count cap has limited T at least once
    kStatCount = 16
};
constexpr int kStatReplicas = 64; // counters are replicated to spread atomic traffic

constexpr int kHistBins = 2048;
constexpr int kAccReplicas = 64; // tighten-pass accumulators (occupied, solid): one 64-byte line per replica

} // namespace mhx
