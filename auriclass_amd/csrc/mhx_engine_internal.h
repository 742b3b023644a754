// mhx_engine_internal.h -- what mhx_engine.cpp (engine state, sketcher, distances) and mhx_files.cpp (file
// ingest and the file-level calls) share.  Internal; the public surface is include/mhx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "mhx_internal.h"

namespace mhx {

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(MHX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// FASTA through the device parser (mhx_files.cpp), kept between files and calls: device buffers sized to the largest file
// seen, one sketcher per (k, s) that is reset between files (no hipMalloc / hipFree and no 200 MB table set-up per file)
constexpr uint32_t kFastaSepsInline = 4096; // record positions that come back with the first synchronisation
struct FastaCtx {
    uint8_t *d_raw[2] = {nullptr, nullptr}; // file i uses d_raw[i & 1]: file i + 1 is copied in while file i is parsed and sketched
    hipEvent_t raw_ready[2] = {nullptr, nullptr};
    uint8_t *d_out = nullptr, *d_ws = nullptr;
    uint64_t *d_seps = nullptr;
    size_t raw_cap = 0, ws_cap = 0;
    uint32_t seps_cap = 0;
    uint64_t *h_words = nullptr; // pinned: [0] stream size, [1] {format flag, #separators}, [2 ..) the first kFastaSepsInline separators
    mhx_sketcher *sk = nullptr;
    int k = 0;
    uint32_t s = 0;
    uint64_t scale = 0;
};

// ---- engine state ---------------------------------------------------------------------
struct Engine {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_dist_ms = 0.0;
    uint8_t *dist_ws = nullptr; // workspace of the all-vs-refs distance path
    size_t dist_ws_cap = 0;
    uint8_t *dist_in = nullptr; // staging of a host-pointer distance batch (rows, lengths, outputs)
    size_t dist_in_cap = 0;
    int last_dist_fallbacks = 0;
    uint8_t *dist_img = nullptr; // pinned image of the reference sketch file of mhx_dist_files
    size_t dist_img_cap = 0;
    // bulk file ingest: pinned staging ring + copy stream (allocated on first use, kept)
    static constexpr int kPinnedSlots = 4;
    uint8_t *pinned[kPinnedSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pinned_free[kPinnedSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    FastaCtx fasta;
    // chunked ingest (.gz FASTQ): pinned host buffers kept between calls, two device slots with their events, a pinned word
    static constexpr size_t kIngestPinnedKeep = 12;
    std::vector<void *> ingest_pinned;
    uint8_t *ingest_slot[2] = {nullptr, nullptr};
    hipEvent_t ingest_copied[2] = {nullptr, nullptr}, ingest_consumed[2] = {nullptr, nullptr};
    uint32_t *ingest_word = nullptr;
};
extern Engine g;
int require_engine();

} // namespace mhx

// FASTQ pushes stay "unsettled" (their bytes may be read again by a repair pass) until a synchronisation point.  A caller
// that recycles its device buffers push by push (the chunked ingest) asks here whether the push that read `d_bytes`, whose
// kernels it knows to have completed, can be let go: the "repair due" word is read on `side` (not behind the kernels of
// later pushes on the engine stream) into the pinned `word`; still zero -> that push needs no repair and is forgotten;
// set -> everything unsettled is repaired now, while all of it is still intact (full synchronisation).  A push that is
// not on the list any more (an earlier repair has taken it) needs nothing.
int sketcher_release_push(mhx_sketcher *sk, const void *d_bytes, hipStream_t side, uint32_t *word);

// sketcher with `table_scale` times the default candidate table and admission budget
int create_sketcher(int k, uint32_t s, uint32_t min_mult, uint64_t expected_bytes, uint64_t table_scale, mhx_sketcher **out);
