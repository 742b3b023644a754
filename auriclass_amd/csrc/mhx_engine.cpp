// mhx_engine.cpp -- host side of libmhx: device selection, the sketcher object that
// schedules tile launches and threshold tightening, the multi-GPU partial export and the
// batched distance entry point.  The file-level calls that replace AuriClass's
// `mash sketch` / `mash dist` subprocesses live in mhx_files.cpp.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>


#include <algorithm>
#include <chrono>
#include <exception>
#include <new>
#include <memory>
#include <string>
#include <vector>

#include "mhx_device.h"
#include "mhx_engine_internal.h"
#include "mhx_internal.h"

namespace mhx {

// ---- errors ---------------------------------------------------------------------------
static thread_local std::string g_err;
int fail(int code, const char *fmt, ...)
{
    char b[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(b, sizeof(b), fmt, ap);
    va_end(ap);
    g_err = b;
    return code;
}
void clear_error() { g_err.clear(); }

Engine g;

int require_engine()
{
    if (!g.ready) return fail(MHX_E_NO_DEVICE, "mhx: no GPU engine (call mhx_init on a machine with a HIP device; there is no CPU fallback)");
    return MHX_OK;
}

} // namespace mhx

using namespace mhx;

extern "C" const char *mhx_last_error(void) { return g_err.c_str(); }
extern "C" const char *mhx_version(void) { return "mhx 0.1.0 (gfx950)"; }
extern "C" void *mhx_stream(void) { return g.stream; }

extern "C" int mhx_init(int device)
{
    clear_error();
    if (g.ready && (device < 0 || device == g.device)) return MHX_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MHX_E_NO_DEVICE, "mhx: no HIP device visible (MI355X required; there is no CPU fallback)");
    if (device < 0) device = 0;
    if (device >= n) return fail(MHX_E_ARG, "mhx_init: device %d out of range (%d visible)", device, n);
    if (g.ready) mhx_shutdown();
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&g.ev0));
    HIPCHK(hipEventCreate(&g.ev1));
    g.device = device;
    g.ready = true;
    return MHX_OK;
}

extern "C" void mhx_shutdown(void)
{
    if (!g.ready) return;
    hipStreamSynchronize(g.stream);
    if (g.fasta.sk) mhx_sketcher_destroy(g.fasta.sk);
    for (int i = 0; i < 2; ++i) { hipFree(g.fasta.d_raw[i]); if (g.fasta.raw_ready[i]) hipEventDestroy(g.fasta.raw_ready[i]); }
    hipFree(g.fasta.d_out); hipFree(g.fasta.d_ws); hipFree(g.fasta.d_seps);
    if (g.fasta.h_words) hipHostFree(g.fasta.h_words);
    hipFree(g.dist_ws);
    if (g.dist_img) hipHostFree(g.dist_img);
    g.dist_img = nullptr;
    g.dist_img_cap = 0;
    hipFree(g.dist_in);
    for (void *p : g.ingest_pinned) hipHostFree(p);
    for (int i = 0; i < 2; ++i) {
        hipFree(g.ingest_slot[i]);
        if (g.ingest_copied[i]) hipEventDestroy(g.ingest_copied[i]);
        if (g.ingest_consumed[i]) hipEventDestroy(g.ingest_consumed[i]);
    }
    if (g.ingest_word) hipHostFree(g.ingest_word);
    if (g.copy_stream) hipStreamSynchronize(g.copy_stream);
    for (int i = 0; i < Engine::kPinnedSlots; ++i) {
        if (g.pinned[i]) hipHostFree(g.pinned[i]);
        if (g.pinned_free[i]) hipEventDestroy(g.pinned_free[i]);
    }
    if (g.copy_stream) hipStreamDestroy(g.copy_stream);
    hipEventDestroy(g.ev0);
    hipEventDestroy(g.ev1);
    hipStreamDestroy(g.stream);
    g = Engine();
}

extern "C" int mhx_device_name(char *buf, size_t cap)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, g.device));
    snprintf(buf, cap, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return MHX_OK;
}

extern "C" int mhx_set_profiling(int on)
{
    g.profiling = on != 0;
    return MHX_OK;
}

// ---- sketcher -------------------------------------------------------------------------
constexpr uint32_t kDeviceOrderMinSketch = 8192;

struct SortScratch {
    std::vector<uint32_t> start;
    std::vector<uint64_t> keys;
    std::vector<uint32_t> cnts;
};

struct mhx_sketcher {
    SortScratch sorted;            // finish() / export(): the extracted entries in hash order
    int k = 0;
    uint32_t s = 0, m = 1;
    bool hash32 = false;
    uint64_t nslots = 0;
    uint64_t hash_max = 0;   // largest representable hash (2^64-1 or 2^32-1)
    uint64_t t_init = 0;     // initial admission threshold (everything admitted)
    // device
    uint64_t *d_keys = nullptr;
    uint32_t *d_cnts = nullptr;
    uint64_t *d_thresh = nullptr;
    uint32_t *d_hist = nullptr;
    uint64_t *d_acc = nullptr;
    uint64_t *d_stats = nullptr;   // kStatReplicas x kStatCount
    uint32_t *d_tickets = nullptr; // one per tile launch since the last reset (kTicketWords of them)
    uint32_t tickets_used = 0;
    uint32_t *d_done = nullptr;    // ticket of the tighten pass
    uint8_t *d_phase_rec = nullptr; // FASTQ: phase_record() per tile of the span being pushed (chain check)
    uint32_t phase_rec_cap = 0;
    uint32_t *d_need = nullptr;    // FASTQ: some tile could not find its line phase by itself -> repair pass due
    struct Span { const void *ptr; uint64_t n; };
    std::vector<Span> unsettled;   // FASTQ pushes whose repair question is still open (their buffers are valid until the next sync)
    uint64_t *d_tile_state = nullptr;
    size_t tile_state_cap = 0;
    uint8_t *d_stage = nullptr;
    size_t stage_cap = 0;
    uint64_t *d_out_keys = nullptr;
    uint32_t *d_out_cnts = nullptr;
    uint32_t *d_out_n = nullptr;
    uint32_t out_cap = 0;
    // sharded path: header of the shard export [n, T, flags, #(2^64-1), occupied, 0, 0, 0], accumulated on the device and
    // handed to the pinned mirror by the extract kernel itself (the entries stay in d_out_keys / d_out_cnts)
    uint64_t *d_exp_hdr = nullptr, *h_exp_hdr = nullptr;
    uint64_t exported = 0;     // entries of the last export_begin (valid until the next push / reset)
    bool export_valid = false;
    bool merged = false;       // merge_slabs has added other shards' entries to the table: reset before the next push
    uint64_t *d_merge_in = nullptr; // staging of gathered slabs that arrive in host memory (gloo)
    size_t merge_in_cap = 0;
    // workspace of the binned merge (mhx_merge.hip): per-bin cursors / counts / flags (kept zero between merges by the
    // kernels), bin regions
    uint32_t *d_mg_small = nullptr;  // [kMergeMaxBins] cursor | [kMergeMaxBins] qn | [16] flags
    uint64_t *d_mg_keys = nullptr;
    uint32_t *d_mg_cnts = nullptr;
    size_t mg_entries = 0;
    // finish(): one device block [n, T, flags, #(2^64-1) | hashes[fin_cap] | counts[fin_cap]] and its pinned host
    // mirror, so the result comes back in ONE copy (five separate copies cost 20-60 us of idle gap each)
    uint64_t *d_fin = nullptr, *h_fin = nullptr;
    uint32_t fin_cap = 0;
    // large sketches: a second block, the first in (almost) hash order (launch_order_block), and its bucket counters
    uint64_t *d_fin_ordered = nullptr;
    uint32_t *d_order_buckets = nullptr, *d_order_starts = nullptr, *d_order_groups = nullptr;
    uint32_t order_log2 = 0;
    bool table_dirty = true;   // tiles have been hashed since the last EXACT tighten pass
    bool table_sampled = false; // ... but a sampled pass has run after the last of them: T is valid and ~s' solid hashes lie below it
    // host
    uint64_t next_chunk_bytes = 0; // geometric schedule of the tightening phase
    uint64_t bytes_pushed = 0;
    uint64_t repair_next_chunk_bytes = 0; // the same two for the FASTQ repair passes: the schedule the left-out tiles would
    uint64_t repair_bytes = 0;            // have had on their own (T is at least as low as that schedule assumes)
    uint64_t expected_bytes = 0;
    uint64_t admit_scale = 1;      // multiplies the initial admission budget (retries after MHX_E_CAPACITY)
    double hash_ms = 0.0;
    uint64_t launches = 0;
    uint64_t last_T = 0;
    // m > 1 only: until s hashes with count >= m exist below T the table is protected by a bound that
    // follows the input seen so far (see push_device)
    bool bounded = false;      // as of the last finish(): the byte-count cap has limited T at least once (m > 1)
    bool established = false;  // as of the last finish(): a tighten pass has lowered T from solid (count >= m) entries
    uint64_t occupied = 0;     // table occupancy reported by the last tighten pass
    uint64_t solid = 0;        // entries <= T with count >= m reported by the last tighten pass
};

static constexpr int kMaxLaunchesPerPush = 64;
static constexpr uint32_t kTicketWords = 4096; // tile launches between two clears of the ticket words
#ifndef MHX_CHUNK_GROWTH
#define MHX_CHUNK_GROWTH 16
#endif
static constexpr uint64_t kChunkGrowth = MHX_CHUNK_GROWTH; // smallest chunk size ratio between tighten rounds
static constexpr uint64_t kUncappedBytes = 1u << 20;       // m > 1: prefix of the input that is admitted whole

static TableArgs table_args(mhx_sketcher *sk)
{
    TableArgs t;
    t.keys = sk->d_keys; t.cnts = sk->d_cnts; t.nslots = sk->nslots; t.thresh = sk->d_thresh;
    t.hist = sk->d_hist; t.acc = sk->d_acc; t.stats = sk->d_stats; t.done = sk->d_done; t.need_lookback = sk->d_need; t.min_mult = sk->m; t.sketch_size = sk->s;
    t.sample = 1;
    t.next_cap = 0;
    return t;
}

static void free_sketcher(mhx_sketcher *sk)
{
    if (!sk) return;
    hipFree(sk->d_keys); hipFree(sk->d_cnts); hipFree(sk->d_thresh); hipFree(sk->d_hist); hipFree(sk->d_acc);
    hipFree(sk->d_stats); hipFree(sk->d_tickets); hipFree(sk->d_done); hipFree(sk->d_need); hipFree(sk->d_phase_rec); hipFree(sk->d_tile_state); hipFree(sk->d_stage);
    hipFree(sk->d_out_keys); hipFree(sk->d_out_cnts); hipFree(sk->d_out_n);
    hipFree(sk->d_exp_hdr); hipFree(sk->d_merge_in); hipFree(sk->d_mg_small); hipFree(sk->d_mg_keys); hipFree(sk->d_mg_cnts);
    if (sk->h_exp_hdr) hipHostFree(sk->h_exp_hdr);
    hipFree(sk->d_fin);
    hipFree(sk->d_fin_ordered);
    hipFree(sk->d_order_buckets);
    hipFree(sk->d_order_starts);
    hipFree(sk->d_order_groups);
    if (sk->h_fin) hipHostFree(sk->h_fin);
    delete sk;
}

static uint64_t next_pow2(uint64_t v)
{
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

extern "C" int mhx_sketcher_reset(mhx_sketcher *sk)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk) return fail(MHX_E_ARG, "null sketcher");
    // Admission threshold: everything is admitted at first.  For m = 1 the first tighten pass already
    // finds s entries; for m > 1 push_device keeps the table safe until s solid hashes exist.
    sk->t_init = sk->hash_max;
    // one launch: table vacated, histogram / accumulators / counters / tickets cleared, T = t_init
    HIPCHK(launch_reset(table_args(sk), sk->t_init, sk->d_tickets, kTicketWords, sk->d_out_n, g.stream));
    sk->tickets_used = 0;
    sk->table_dirty = true;
    sk->merged = false;
    sk->export_valid = false;
    sk->unsettled.clear();
    sk->last_T = sk->hash_max;
    sk->bounded = false;
    sk->established = false;
    sk->occupied = 0;
    sk->solid = 0;
    uint64_t c0 = next_pow2((uint64_t)sk->s * 64);
    if (c0 < (1u << 20)) c0 = 1u << 20;
    if (sk->m > 1) c0 = kUncappedBytes; // multiplicity filter: the first stage is admitted whole (see push_device)
    if (c0 > sk->nslots / 4) c0 = sk->nslots / 4; // first chunk may admit every position
    sk->next_chunk_bytes = c0;
    sk->bytes_pushed = 0;
    sk->repair_next_chunk_bytes = c0;
    sk->repair_bytes = 0;
    sk->hash_ms = 0.0;
    sk->launches = 0;
    return MHX_OK;
}

int create_sketcher(int k, uint32_t s, uint32_t min_mult, uint64_t expected_bytes, uint64_t table_scale, mhx_sketcher **out)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!out) return fail(MHX_E_ARG, "null out pointer");
    if (!hash_k_supported(k)) return fail(MHX_E_ARG, "k-mer size %d not supported (1..32)", k);
    if (s == 0) return fail(MHX_E_ARG, "sketch size must be positive");
    mhx_sketcher *sk = new mhx_sketcher();
    sk->k = k; sk->s = s; sk->m = min_mult ? min_mult : 1;
    sk->hash32 = k <= 16;
    sk->hash_max = sk->hash32 ? 0xFFFFFFFFull : ~0ull;
    sk->expected_bytes = expected_bytes;
    sk->admit_scale = table_scale ? table_scale : 1;
    // table: >= 2^21 slots, >= 256 slots per sketch entry (worst-case admissions of the
    // occurrence bound at load 1; real inputs repeat their k-mers and stay far below)
    // (round 3: at least 2^21 slots instead of 2^22 -- the first chunk and every later one are sized as fractions of the
    // table, so the load bounds are the same, and a table half the size is reset, histogrammed and extracted in half the
    // time: +1.4 % on the headline configuration.  128 slots per entry would buy AuriClass's defaults another 0.5 % but
    // puts the worst case (every k-mer distinct) at 60 % load; 64: slower, the probe sequences get long.)
    static const uint64_t slots_per_entry = getenv("MHX_SLOTS_PER_ENTRY") ? (uint64_t)atol(getenv("MHX_SLOTS_PER_ENTRY")) : 256;
    static const int min_log2 = getenv("MHX_TABLE_MIN_LOG2") ? atoi(getenv("MHX_TABLE_MIN_LOG2")) : 21; // experiment knobs
    uint64_t want = (uint64_t)s * slots_per_entry;
    if (want < (1ull << min_log2)) want = 1ull << min_log2;
    if (expected_bytes && expected_bytes * 4 < want && expected_bytes * 4 >= (1ull << 16)) want = expected_bytes * 4;
    if (expected_bytes && expected_bytes * 4 < (1ull << 16)) want = 1ull << 16;
    want *= table_scale;
    if (want > (1ull << 30)) want = 1ull << 30; // 12.9 GB of table at most
    sk->nslots = next_pow2(want);
    sk->out_cap = s * 2 + 65536;
    hipError_t e = hipSuccess;
    auto A = [&](void **p, size_t n) { if (e == hipSuccess) e = hipMalloc(p, n); };
    A((void **)&sk->d_keys, sk->nslots * sizeof(uint64_t));
    A((void **)&sk->d_cnts, sk->nslots * sizeof(uint32_t));
    A((void **)&sk->d_thresh, sizeof(uint64_t));
    A((void **)&sk->d_hist, kHistBins * sizeof(uint32_t));
    A((void **)&sk->d_acc, kAccReplicas * 8 * sizeof(uint64_t));
    A((void **)&sk->d_stats, kStatReplicas * kStatCount * sizeof(uint64_t));
    A((void **)&sk->d_tickets, kTicketWords * sizeof(uint32_t));
    A((void **)&sk->d_done, sizeof(uint32_t));
    A((void **)&sk->d_need, sizeof(uint32_t));
    A((void **)&sk->d_out_keys, sk->out_cap * sizeof(uint64_t));
    A((void **)&sk->d_out_cnts, sk->out_cap * sizeof(uint32_t));
    A((void **)&sk->d_out_n, sizeof(uint32_t));
    A((void **)&sk->d_exp_hdr, 8 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemset(sk->d_exp_hdr, 0, 8 * sizeof(uint64_t)); // zero between two exports (the kernel clears them)
    if (e == hipSuccess) e = hipHostMalloc((void **)&sk->h_exp_hdr, 8 * sizeof(uint64_t), hipHostMallocDefault);
    sk->fin_cap = (s + 16u * (uint32_t)sqrt((double)s) + 4096u + 1u) & ~1u; // what a sampled threshold leaves, with room (finish())
    const size_t fin_bytes = (4 + (size_t)sk->fin_cap + sk->fin_cap / 2) * sizeof(uint64_t);
    A((void **)&sk->d_fin, fin_bytes);
    if (e == hipSuccess) e = hipMemset(sk->d_fin, 0, 4 * sizeof(uint64_t)); // the header words are zero between two finish() calls
    if (s >= kDeviceOrderMinSketch) { // below that the host's bucket sort costs less than three more launches
        sk->order_log2 = 12;
        while ((1u << sk->order_log2) < sk->fin_cap && sk->order_log2 < 20) ++sk->order_log2;
        A((void **)&sk->d_fin_ordered, fin_bytes);
        A((void **)&sk->d_order_buckets, ((size_t)1 << sk->order_log2) * sizeof(uint32_t));
        A((void **)&sk->d_order_starts, (((size_t)1 << sk->order_log2) + 4) * sizeof(uint32_t));
        A((void **)&sk->d_order_groups, 1024 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(sk->d_order_starts, 0, (((size_t)1 << sk->order_log2) + 4) * sizeof(uint32_t)); // [nbuckets] stays zero
        if (e == hipSuccess) e = hipMemset(sk->d_order_buckets, 0, ((size_t)1 << sk->order_log2) * sizeof(uint32_t)); // every finish() leaves them zero again
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&sk->h_fin, fin_bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        free_sketcher(sk);
        return fail(MHX_E_HIP, "hipMalloc failed while creating the sketcher: %s", hipGetErrorString(e));
    }
    rc = mhx_sketcher_reset(sk);
    if (rc) { free_sketcher(sk); return rc; }
    *out = sk;
    return MHX_OK;
}

extern "C" int mhx_sketcher_create(int k, uint32_t s, uint32_t min_mult, uint64_t expected_bytes, mhx_sketcher **out)
{
    return create_sketcher(k, s, min_mult, expected_bytes, 1, out);
}

extern "C" int mhx_sketcher_create_scaled(int k, uint32_t s, uint32_t min_mult, uint64_t expected_bytes, uint32_t budget_scale, mhx_sketcher **out)
{
    return create_sketcher(k, s, min_mult, expected_bytes, budget_scale ? budget_scale : 1, out);
}

extern "C" void mhx_sketcher_destroy(mhx_sketcher *sk)
{
    if (g.ready) hipStreamSynchronize(g.stream);
    free_sketcher(sk);
}

static int read_threshold(mhx_sketcher *sk, uint64_t *T)
{ // after a tighten pass: the threshold and the table occupancy that pass measured
    static_assert(kStatSolid == kStatOccupied + 1, "occupied and solid are read with one copy");
    uint64_t occ_solid[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(T, sk->d_thresh, sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(occ_solid, sk->d_stats + kStatOccupied, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    sk->last_T = *T;
    sk->occupied = occ_solid[0];
    sk->solid = occ_solid[1];
    return MHX_OK;
}

// FASTQ runs in two kernel forms (mhx_kernels.hip): kernel format 2, every tile finds its line phase by itself -- no
// ticket, no wait between workgroups --, and format 1, ticket + decoupled look-back.  A push goes through format 2; tiles
// whose lines are too long to self-synchronise (reads beyond ~2.7 kb) leave themselves out and raise a word that
// settle() reads at the next synchronisation point; the repair pass then runs format 1 over the same span with only
// those tiles doing work.  (MHX_NO_SELFSYNC=1: format 1 for everything, as in round 1.)
static int push_span(mhx_sketcher *sk, const void *d_bytes, uint64_t n, int kfmt, bool repair);

static int repair_unsettled(mhx_sketcher *sk)
{
    std::vector<mhx_sketcher::Span> spans;
    spans.swap(sk->unsettled);
    HIPCHK(hipMemsetAsync(sk->d_need, 0, sizeof(uint32_t), g.stream));
    for (const auto &sp : spans) {
        const int rc = push_span(sk, sp.ptr, sp.n, 1, true);
        if (rc) return rc;
    }
    return MHX_OK;
}

// at a synchronisation point that is not finish(): is a repair pass due for the pushes since the last one?
static int settle(mhx_sketcher *sk)
{
    if (sk->unsettled.empty()) return MHX_OK;
    uint32_t *need = reinterpret_cast<uint32_t *>(sk->h_fin); // pinned landing word
    HIPCHK(hipMemcpyAsync(need, sk->d_need, sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    if (*need) {
        const int rc = repair_unsettled(sk);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    sk->unsettled.clear();
    return MHX_OK;
}

int sketcher_release_push(mhx_sketcher *sk, const void *d_bytes, hipStream_t side, uint32_t *word)
{
    if (!sk) return MHX_OK;
    size_t at = sk->unsettled.size();
    for (size_t i = 0; i < sk->unsettled.size(); ++i)
        if (sk->unsettled[i].ptr == d_bytes) { at = i; break; }
    if (at == sk->unsettled.size()) return MHX_OK; // settled already (an earlier repair pass took everything there was)
    HIPCHK(hipMemcpyAsync(word, sk->d_need, sizeof(uint32_t), hipMemcpyDeviceToHost, side));
    HIPCHK(hipStreamSynchronize(side));
    if (*word == 0) {
        sk->unsettled.erase(sk->unsettled.begin() + (long)at);
        return MHX_OK;
    }
    int rc = repair_unsettled(sk); // reads every unsettled span again: all of them are still in place
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    return MHX_OK;
}

extern "C" int mhx_sketcher_push_device(mhx_sketcher *sk, const void *d_bytes, uint64_t n, int fmt)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || (!d_bytes && n)) return fail(MHX_E_ARG, "null argument");
    if (fmt != MHX_FMT_SEQ && fmt != MHX_FMT_FASTQ4) return fail(MHX_E_ARG, "unknown stream format %d", fmt);
    if (sk->merged) return fail(MHX_E_ARG, "this sketcher holds a merged (multi-shard) table: mhx_sketcher_reset() before the next push");
    if (n == 0) return MHX_OK;
    sk->export_valid = false;
    if (fmt == MHX_FMT_SEQ) return push_span(sk, d_bytes, n, 0, false);
    static const bool no_selfsync = getenv("MHX_NO_SELFSYNC") != nullptr;
    if (no_selfsync) return push_span(sk, d_bytes, n, 1, false);
    if (sk->unsettled.size() >= 4096) { // thousands of small pushes without a synchronisation point: settle what there is
        rc = settle(sk);
        if (rc) return rc;
    }
    rc = push_span(sk, d_bytes, n, 2, false);
    if (!rc) sk->unsettled.push_back({d_bytes, n});
    return rc;
}

static int push_span(mhx_sketcher *sk, const void *d_bytes, uint64_t n, int kfmt, bool repair)
{
    const uintptr_t p = (uintptr_t)d_bytes;
    const uintptr_t base = p & ~(uintptr_t)15;
    HashArgs a;
    a.base = (const uint8_t *)base;
    a.begin = p - base;
    a.end = a.begin + n;
    a.first_tile = 0;
    a.hash32 = sk->hash32 ? 1 : 0;
    a.thresh = sk->d_thresh;
    a.keys = sk->d_keys; a.cnts = sk->d_cnts; a.slot_mask = sk->nslots - 1; a.stats = sk->d_stats;
    a.need_lookback = sk->d_need;
    a.repair = repair ? 1u : 0u;
    static const char *force_queue = getenv("MHX_QUEUE_CANDIDATES"); // "0" / "1": diagnostic override
    a.queue_candidates = force_queue ? (uint32_t)(force_queue[0] == '1') : (uint32_t)(sk->s >= kDeviceOrderMinSketch);
    const uint64_t ntiles64 = (a.end + kTileBytes - 1) / kTileBytes;
    if (ntiles64 > 0x7FFFFFFFull) return fail(MHX_E_ARG, "span too large for one push (%llu bytes)", (unsigned long long)n);
    const uint32_t ntiles = (uint32_t)ntiles64;
    if (kfmt == 1) {
        if (sk->tile_state_cap < ntiles) {
            HIPCHK(hipStreamSynchronize(g.stream));
            hipFree(sk->d_tile_state);
            sk->d_tile_state = nullptr;
            sk->tile_state_cap = 0;
            HIPCHK(hipMalloc((void **)&sk->d_tile_state, (size_t)ntiles * sizeof(uint64_t)));
            sk->tile_state_cap = ntiles;
        }
        HIPCHK(hipMemsetAsync(sk->d_tile_state, 0, (size_t)ntiles * sizeof(uint64_t), g.stream));
        if (sk->tickets_used + kMaxLaunchesPerPush > kTicketWords) { // every launch takes a fresh, still zero word
            HIPCHK(hipMemsetAsync(sk->d_tickets, 0, kTicketWords * sizeof(uint32_t), g.stream));
            sk->tickets_used = 0;
        }
    }
    a.tile_state = sk->d_tile_state;
    if (kfmt == 2 || repair) { // every tile of the span writes its record: nothing to clear
        if (sk->phase_rec_cap < ntiles) {
            HIPCHK(hipStreamSynchronize(g.stream));
            hipFree(sk->d_phase_rec);
            sk->d_phase_rec = nullptr;
            sk->phase_rec_cap = 0;
            HIPCHK(hipMalloc((void **)&sk->d_phase_rec, (size_t)ntiles));
            sk->phase_rec_cap = ntiles;
        }
    }
    a.phase_rec = sk->d_phase_rec;
    TableArgs ta = table_args(sk);
    if (sk->nslots >= (1ull << 23) && !getenv("MHX_EXACT_TIGHTEN")) ta.sample = 8; // big tables: sampled passes between chunks (finish() counts exactly)
    // a repair pass runs the same staged schedule on counters of its own (it may be the first time any k-mer is admitted)
    uint64_t &next_chunk_bytes = repair ? sk->repair_next_chunk_bytes : sk->next_chunk_bytes;
    uint64_t &bytes_pushed = repair ? sk->repair_bytes : sk->bytes_pushed;
    const uint64_t pushed_before = bytes_pushed;
    const bool filtered = sk->m > 1;
    // The host never looks at T while pushing: every launch is followed by a tighten pass on the stream and nothing
    // waits for a round trip.
    // No multiplicity filter: launches grow by a factor G.  After a chunk of N k-mers T sits at the s-th smallest of
    // them, hence the next, G times larger chunk admits ~G*s occurrences: G is what keeps that at a sixteenth of the
    // table whatever the input is.
    // Multiplicity filter (m > 1): T cannot follow the data before s hashes with count >= m exist, and until then
    // every admitted k-mer costs two atomics and may be a new table entry.  The first MiB is admitted whole (small
    // genomes and saturated k-mer spaces show their solid hashes there); after that the bytes seen grow x8 (m <= 3) or x4 per launch
    // and, in front of every launch, T is capped ON THE DEVICE at 48*s' / (bytes seen after this launch),
    // s' = s + 8*sqrt(s) + 16, i.e. ~20*s admissions per stage -- unless a tighten pass has meanwhile lowered T from
    // solid hashes, or the table looks like a small genome sequenced deeply (cap_threshold_kernel; inside a push the
    // tighten pass in front of the launch applies the cap itself, TableArgs::next_cap).  The cap stays
    // above the final s-th solid hash for any genome size while the error-free k-mer coverage c so far is <= ~17x,
    // and s solid hashes appear below it as soon as c / P[Poisson(c) >= m] <= 17 (c in 0.8 .. 16 for m = 3), a window
    // no x4 stage can jump over.  Inputs with fewer than s solid k-mers in total, or m > ~8, end in finish()'s
    // exactness check and the retry with a 16x budget.
    struct Plan { uint32_t take; uint64_t cap; };
    auto plan = [&](uint32_t tile, int launch) {
        Plan p{ntiles - tile, 0};
        if (launch != kMaxLaunchesPerPush - 1) {
            uint64_t chunk_bytes = next_chunk_bytes;
            if (filtered) {
                // stages are defined on the bytes actually seen (pushes may be of any size): the uncapped first MiB,
                // then never more than x4 (x8 for m <= 3) cumulative growth per launch
                // (x8 for m <= 3, round 3: the byte-count cap admits ~19 s' (1 - 1/g) occurrences per stage whatever the growth g
                // is, and the window of coverages in which s solid hashes lie below the cap -- c / P[Poisson(c) >= m] <= 17:
                // c in 0.8 .. 16 for m = 3, 1.6 .. 16 for m = 4 -- spans a factor 20 resp. 10: no x8 stage can jump over it.
                // Two launches and two passes fewer on a 3 GB input.  Larger m keep x4: 2.5 .. 16 for m = 5.)
                const uint64_t rest_of_prefix = bytes_pushed < kUncappedBytes ? kUncappedBytes - bytes_pushed : 0;
                chunk_bytes = std::max<uint64_t>(rest_of_prefix, (sk->m <= 3 ? 7 : 3) * bytes_pushed);
            }
            const uint64_t chunk_tiles = std::max<uint64_t>(1, chunk_bytes / kTileBytes);
            if (chunk_tiles < p.take) p.take = (uint32_t)chunk_tiles;
        }
        if (filtered) {
            const uint64_t after = pushed_before + std::min<uint64_t>(n, (uint64_t)(tile + p.take) * kTileBytes);
            if (after > kUncappedBytes) {
                const long double s_eff = (long double)sk->s + 8.0L * sqrtl((long double)sk->s) + 16.0L;
                const long double cap_frac = 48.0L * s_eff * (long double)sk->admit_scale / (long double)after;
                if (cap_frac < 1.0L) p.cap = std::max<uint64_t>(1, (uint64_t)(cap_frac * (long double)sk->hash_max));
            }
        }
        return p;
    };
    uint32_t tile = 0;
    int launch = 0;
    Plan cur = plan(0, 0);
    if (cur.cap) HIPCHK(launch_cap_threshold(sk->d_thresh, cur.cap, sk->d_stats, g.stream)); // first launch of a push: a launch of its own
    while (tile < ntiles) {
        // Kernel form of this launch (process_group_regs): candidates are queued when many windows will pass the admission
        // test -- a large sketch, or an early launch whose threshold still stems from little data (T ~ s-th smallest of
        // the k-mers seen so far, ~0.4 per FASTQ byte: above ~3 candidates in 10^4 windows the queue wins); the very first
        // launch, which admits everything, and the long launches of a small sketch finish them where they are found.
        if (!force_queue) {
            // (nothing pushed yet: T is still at its initial value and admits everything)
            long double expected_rate = bytes_pushed ? std::min(1.0L, (long double)sk->s / (0.4L * (long double)bytes_pushed)) : 1.0L;
            // staged phase of the multiplicity filter: the threshold sits at the byte-count cap until solid hashes take over
            if (cur.cap) expected_rate = std::max(expected_rate, (long double)cur.cap / (long double)sk->hash_max);
            // (a sequence stream fills the work list -- every group of a tile is an item --, which leaves the queue no room;
            // above ~0.15 candidates per window the ~1100 free entries of a FASTQ tile's list overflow and the tile would do
            // its work twice, see sketch_tile_kernel: such launches finish their candidates inline)
            a.queue_candidates = (uint32_t)(kfmt != 0 && expected_rate <= 0.1L && (sk->s >= kDeviceOrderMinSketch || expected_rate > 3e-4L));
        }
        a.tile0 = tile;
        a.ntiles = cur.take;
        a.ticket = sk->d_tickets + sk->tickets_used++;
        if (g.profiling) HIPCHK(hipEventRecord(g.ev0, g.stream));
        HIPCHK(launch_hash(sk->k, kfmt, a, g.stream));
        if (g.profiling) {
            HIPCHK(hipEventRecord(g.ev1, g.stream));
            HIPCHK(hipEventSynchronize(g.ev1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, g.ev0, g.ev1));
            sk->hash_ms += ms;
        }
        ++sk->launches;
        ++launch;
        tile += cur.take;
        sk->table_dirty = true;
        sk->table_sampled = false;
        bytes_pushed = pushed_before + std::min<uint64_t>(n, (uint64_t)tile * kTileBytes); // real bytes, not whole tiles: callers may push tiny spans
        if (!filtered && next_chunk_bytes < (1ull << 40)) {
            uint64_t G = sk->nslots / (16ull * sk->s);
            G = std::min<uint64_t>(std::max<uint64_t>(G, kChunkGrowth), 256);
            next_chunk_bytes *= G;
        }
        if (tile < ntiles) cur = plan(tile, launch);
        // tighten T from what has been seen (also after the last launch of a push: the next push starts from it).
        // Sampled passes (big tables) leave the table marked dirty (an exact pass has not seen it) but sampled: see finish().
        // (Tried in round 3: the passes between two launches on a side stream, beside the next launch.  Exactness survives
        // -- any earlier T is a valid bound -- but capacity does not: the next launch's first workgroups then run with the
        // threshold of TWO chunks ago, which for the second launch is "admit everything" and overflows the table, and
        // with a multiplicity filter the cap in front of the next launch is decided before the pass can report solid
        // hashes.  What remains safe saves < 1 % of a step; the passes stay in stream order.)
        ta.next_cap = tile < ntiles ? cur.cap : 0;
        HIPCHK(launch_tighten(ta, g.stream));
        if (ta.sample == 1) sk->table_dirty = false;
        else sk->table_sampled = true;
    }
    if (kfmt == 2 || repair) HIPCHK(launch_phase_verify(sk->d_phase_rec, ntiles, sk->d_stats, g.stream));
    return MHX_OK;
}

extern "C" int mhx_sketcher_push_host(mhx_sketcher *sk, const void *h_bytes, uint64_t n, int fmt)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || (!h_bytes && n)) return fail(MHX_E_ARG, "null argument");
    if (n == 0) return MHX_OK;
    // the staging buffer is reused: earlier pushes that may still read it (or need it for a repair pass) come first
    rc = settle(sk);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    if (sk->stage_cap < n + 64) {
        hipFree(sk->d_stage);
        sk->d_stage = nullptr;
        sk->stage_cap = 0;
        const size_t cap = (size_t)((n + 64 + (1u << 20) - 1) & ~(uint64_t)((1u << 20) - 1));
        HIPCHK(hipMalloc((void **)&sk->d_stage, cap));
        sk->stage_cap = cap;
    }
    HIPCHK(hipMemcpyAsync(sk->d_stage, h_bytes, n, hipMemcpyHostToDevice, g.stream));
    return mhx_sketcher_push_device(sk, sk->d_stage, n, fmt);
}

extern "C" int mhx_sketcher_sync(mhx_sketcher *sk)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (sk) {
        rc = settle(sk); // a repair pass, if one is due, runs while the pushed buffers are still the caller's to keep
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    return MHX_OK;
}

static int fetch_stats(mhx_sketcher *sk, uint64_t *sum)
{
    std::vector<uint64_t> h(kStatReplicas * kStatCount);
    HIPCHK(hipMemcpyAsync(h.data(), sk->d_stats, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int i = 0; i < kStatCount; ++i) sum[i] = 0;
    for (int r = 0; r < kStatReplicas; ++r) {
        sum[kStatKmers] += h[r * kStatCount + kStatKmers];
        sum[kStatInserts] += h[r * kStatCount + kStatInserts];
        sum[kStatLines] += h[r * kStatCount + kStatLines];
        sum[kStatRecords] += h[r * kStatCount + kStatRecords];
        sum[kStatMaxKey] += h[r * kStatCount + kStatMaxKey];
        sum[kStatFlags] |= h[r * kStatCount + kStatFlags];
    }
    sum[kStatOccupied] = h[kStatOccupied];
    sum[kStatSolid] = h[kStatSolid];
    return MHX_OK;
}

extern "C" int mhx_sketcher_stats(mhx_sketcher *sk, uint64_t *stats8)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !stats8) return fail(MHX_E_ARG, "null argument");
    rc = settle(sk);
    if (rc) return rc;
    uint64_t s[kStatCount];
    rc = fetch_stats(sk, s);
    if (rc) return rc;
    stats8[0] = s[kStatKmers];
    stats8[1] = s[kStatInserts];
    stats8[2] = s[kStatLines];
    stats8[3] = s[kStatFlags];
    stats8[4] = s[kStatOccupied];
    double ms = sk->hash_ms;
    memcpy(&stats8[5], &ms, sizeof(double));
    stats8[6] = sk->launches;
    stats8[7] = sk->last_T;
    return MHX_OK;
}

// diagnostic: raw per-phase cycle sums of a -DMHX_STAMPS build (zeros otherwise)
extern "C" int mhx_sketcher_record_count(mhx_sketcher *sk, uint64_t *records)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !records) return fail(MHX_E_ARG, "null argument");
    rc = settle(sk);
    if (rc) return rc;
    uint64_t s[kStatCount];
    rc = fetch_stats(sk, s);
    if (rc) return rc;
    *records = s[kStatRecords];
    return MHX_OK;
}

extern "C" int mhx_sketcher_debug_stamps(mhx_sketcher *sk, uint64_t *out8)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !out8) return fail(MHX_E_ARG, "null argument");
    std::vector<uint64_t> h(kStatReplicas * kStatCount);
    HIPCHK(hipMemcpyAsync(h.data(), sk->d_stats, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int i = 0; i < 8; ++i) {
        out8[i] = 0;
        for (int r = 0; r < kStatReplicas; ++r) out8[i] += h[r * kStatCount + kStatStamp0 + i];
    }
    return MHX_OK;
}

extern "C" int mhx_sketcher_threshold(mhx_sketcher *sk, uint64_t *threshold)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !threshold) return fail(MHX_E_ARG, "null argument");
    rc = settle(sk);
    if (rc) return rc;
    HIPCHK(launch_tighten(table_args(sk), g.stream));
    return read_threshold(sk, threshold);
}

static int check_flags(uint64_t flags)
{
    if (flags & kFlagSpinTimeout) return fail(MHX_E_INTERNAL, "device look-back timed out");
    if (flags & kFlagTableFull) return fail(MHX_E_CAPACITY, "device candidate table overflowed; recreate the sketcher with a larger expected_bytes");
    if (flags & kFlagBadFastq) return fail(MHX_E_FORMAT, "input is not strict 4-line FASTQ (use the record parser path)");
    return MHX_OK;
}

// entries (key <= limit, count >= min_count) -> host vectors, unsorted
static int extract(mhx_sketcher *sk, uint64_t limit, uint32_t min_count, std::vector<uint64_t> &keys, std::vector<uint32_t> &cnts)
{
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIPCHK(hipMemsetAsync(sk->d_out_n, 0, sizeof(uint32_t), g.stream));
        HIPCHK(launch_extract(table_args(sk), limit, min_count, sk->d_out_keys, sk->d_out_cnts, sk->out_cap, sk->d_out_n, nullptr, nullptr, nullptr, nullptr, g.stream));
        uint32_t n = 0;
        HIPCHK(hipMemcpyAsync(&n, sk->d_out_n, sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        if (n > sk->out_cap) { // grow once and repeat
            hipFree(sk->d_out_keys); hipFree(sk->d_out_cnts);
            sk->d_out_keys = nullptr; sk->d_out_cnts = nullptr;
            sk->out_cap = n + 1024;
            HIPCHK(hipMalloc((void **)&sk->d_out_keys, (size_t)sk->out_cap * sizeof(uint64_t)));
            HIPCHK(hipMalloc((void **)&sk->d_out_cnts, (size_t)sk->out_cap * sizeof(uint32_t)));
            continue;
        }
        keys.resize(n);
        cnts.resize(n);
        if (n) {
            HIPCHK(hipMemcpyAsync(keys.data(), sk->d_out_keys, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipMemcpyAsync(cnts.data(), sk->d_out_cnts, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipStreamSynchronize(g.stream));
        }
        return MHX_OK;
    }
    return fail(MHX_E_INTERNAL, "extract: output kept growing");
}

static void insertion_pass(uint64_t *k2, uint32_t *c2, size_t n)
{ // O(n + inversions): what is left to do after a scatter by leading bits
    for (size_t i = 1; i < n; ++i) {
        const uint64_t k = k2[i];
        if (k2[i - 1] <= k) continue;
        const uint32_t c = c2[i];
        size_t j = i;
        while (j > 0 && k2[j - 1] > k) { k2[j] = k2[j - 1]; c2[j] = c2[j - 1]; --j; }
        k2[j] = k;
        c2[j] = c;
    }
}

static void sort_pairs(const uint64_t *keys, const uint32_t *cnts, size_t n, SortScratch &sc)
{ // The extracted hashes are (close to) uniform below the threshold: one scatter into ~n..2n buckets by their leading
  // bits (a shift, no division), then an insertion sort over the almost-sorted result (O(n) expected; any input still
  // ends up sorted).  Result in sc.keys / sc.cnts; the scratch vectors live with the sketcher (no page faults per call).
    sc.keys.resize(n);
    sc.cnts.resize(n);
    if (n == 0) return;
    uint64_t hi = 0;
    for (size_t i = 0; i < n; ++i) hi = keys[i] > hi ? keys[i] : hi;
    int shift = 0;
    while ((hi >> shift) >= 2 * n) ++shift;
    const size_t nb = (size_t)(hi >> shift) + 1;
    sc.start.assign(nb + 1, 0);
    uint32_t *start = sc.start.data();
    for (size_t i = 0; i < n; ++i) ++start[(size_t)(keys[i] >> shift) + 1];
    for (size_t b = 0; b < nb; ++b) start[b + 1] += start[b];
    uint64_t *k2 = sc.keys.data();
    uint32_t *c2 = sc.cnts.data();
    for (size_t i = 0; i < n; ++i) {
        const size_t d = start[(size_t)(keys[i] >> shift)]++;
        k2[d] = keys[i];
        c2[d] = cnts[i];
    }
    insertion_pass(k2, c2, n);
}

static int mhx_sketcher_finish_impl(mhx_sketcher *sk, uint64_t *hashes, uint32_t *counts, uint32_t *n_out)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !hashes || !n_out) return fail(MHX_E_ARG, "null argument");
    // One batch on the stream, one copy, one synchronisation: final (exact) tighten unless the last pass already was
    // one, extract with the threshold read on the device into the result block, the block to its pinned mirror.
    const uint32_t cap = sk->fin_cap;
    uint64_t *d = sk->d_fin;
    const size_t fin_bytes = (4 + (size_t)cap + cap / 2) * sizeof(uint64_t);
    static const bool dbg = getenv("MHX_FINISH_DEBUG") != nullptr; // stderr: where finish() spends its time
    const auto t_begin = std::chrono::steady_clock::now();
    bool want_exact = false;
  again:
    // big tables: the sampled pass behind the last launch left T at about the (s + 8 sqrt(s))-th solid hash, the block holds
    // twice that, and the host keeps the first s -- a second pass over the table only if that turns out not to be so
    if (sk->table_dirty && (want_exact || !sk->table_sampled)) {
        HIPCHK(launch_tighten(table_args(sk), g.stream));
        sk->table_dirty = false;
        sk->table_sampled = false;
    }
    const bool ordered = sk->d_fin_ordered != nullptr;
    if (ordered) { // large sketches: the kernels put the block in hash order and store it into the pinned block themselves
        HIPCHK(hipMemsetAsync(d, 0, 4 * sizeof(uint64_t), g.stream));
        HIPCHK(launch_extract(table_args(sk), 0, sk->m, d + 4, (uint32_t *)(d + 4 + cap), cap, (uint32_t *)d, d + 2, sk->d_thresh, d + 1, d + 3, g.stream,
                              sk->d_order_buckets, sk->order_log2));
        HIPCHK(launch_order_block(d, cap, sk->order_log2, sk->d_order_buckets, sk->d_order_starts, sk->d_order_groups, sk->d_fin_ordered, sk->h_fin, g.stream));
    } else { // one kernel, nothing else: the entries go straight into the pinned block, the workgroup that finishes last
             // adds the header words (kept on the device while they are being accumulated) and clears them for the next call
        HIPCHK(launch_extract(table_args(sk), 0, sk->m, sk->h_fin + 4, (uint32_t *)(sk->h_fin + 4 + cap), cap, (uint32_t *)d, d + 2, sk->d_thresh, d + 1, d + 3,
                              g.stream, nullptr, 0, d, sk->h_fin, sk->d_done));
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    const auto t_device = std::chrono::steady_clock::now();
    const uint64_t *h = sk->h_fin;
    const uint32_t n = (uint32_t)h[0];
    const uint64_t T = h[1], flags = h[2], maxkey = h[3];
    if ((flags & kFlagNeedLookback) && !sk->unsettled.empty()) { // FASTQ tiles left out by the self-synchronising pass
        rc = repair_unsettled(sk);
        if (rc) return rc;
        goto again;
    }
    sk->unsettled.clear();
    if (n > cap && sk->table_dirty && !want_exact) { // the sampled threshold was not the expected one: count exactly
        want_exact = true;
        goto again;
    }
    sk->last_T = T;
    sk->bounded = (flags & kFlagStateBounded) != 0;
    sk->established = (flags & kFlagStateEstablished) != 0;
    rc = check_flags(flags);
    if (rc) return rc;
    std::vector<uint64_t> keys;
    std::vector<uint32_t> cnts;
    const uint64_t *src_keys = h + 4; // the common case: sorted straight out of the pinned block
    const uint32_t *src_cnts = reinterpret_cast<const uint32_t *>(h + 4 + cap);
    size_t n_src = n;
    const bool extra = T == ~0ull && maxkey >= sk->m; // the one hash value the table cannot hold
    if (n > cap || extra) {
        if (n > cap) { // more entries below T than the result block holds: the general path
            rc = extract(sk, T, sk->m, keys, cnts);
            if (rc) return rc;
        } else {
            keys.assign(src_keys, src_keys + n);
            cnts.assign(src_cnts, src_cnts + n);
        }
        if (extra) {
            keys.push_back(~0ull);
            cnts.push_back((uint32_t)maxkey);
        }
        src_keys = keys.data();
        src_cnts = cnts.data();
        n_src = keys.size();
    }
    // exactness: either nothing was ever rejected (T still at its initial value), or at least s qualifying
    // hashes lie below T.  Fewer than s below a lowered T means the bound was too tight: a host-imposed cap
    // of the m > 1 phase (sk->bounded), or -- never seen, ~1e-9 per pass -- a sampled tighten pass that overshot.
    if (n_src < sk->s && T < sk->hash_max)
        return fail(MHX_E_CAPACITY, "admission threshold was too tight for this input (%zu of %u sketch entries%s); recreate the sketcher with a larger table",
                    n_src, sk->s, sk->bounded ? ", capped threshold" : "");
    const auto t_copy = std::chrono::steady_clock::now();
    const uint32_t nn = n_src < sk->s ? (uint32_t)n_src : sk->s;
    bool in_order = ordered && src_keys == h + 4; // what the ordering kernels promise, checked
    for (size_t i = 1; in_order && i < n_src; ++i) in_order = src_keys[i - 1] < src_keys[i];
    if (in_order) {
        memcpy(hashes, src_keys, (size_t)nn * sizeof(uint64_t));
        if (counts) memcpy(counts, src_cnts, (size_t)nn * sizeof(uint32_t));
    } else {
        sort_pairs(src_keys, src_cnts, n_src, sk->sorted);
        memcpy(hashes, sk->sorted.keys.data(), (size_t)nn * sizeof(uint64_t));
        if (counts) memcpy(counts, sk->sorted.cnts.data(), (size_t)nn * sizeof(uint32_t));
    }
    *n_out = nn;
    if (dbg) {
        const auto t_end = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[mhx finish] device+copy %.0f us (block %zu bytes, %u entries), unpack %.0f us, sort+out %.0f us\n",
                us(t_begin, t_device), fin_bytes, n, us(t_device, t_copy), us(t_copy, t_end));
    }
    return MHX_OK;
}

static int mhx_sketcher_export_impl(mhx_sketcher *sk, uint64_t limit, uint64_t *hashes, uint32_t *counts, uint32_t cap, uint32_t *n_out)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !n_out) return fail(MHX_E_ARG, "null argument");
    rc = settle(sk);
    if (rc) return rc;
    uint64_t st[kStatCount];
    rc = fetch_stats(sk, st);
    if (rc) return rc;
    rc = check_flags(st[kStatFlags]);
    if (rc) return rc;
    std::vector<uint64_t> keys;
    std::vector<uint32_t> cnts;
    rc = extract(sk, limit, 1, keys, cnts);
    if (rc) return rc;
    if (limit == ~0ull && st[kStatMaxKey]) { keys.push_back(~0ull); cnts.push_back((uint32_t)st[kStatMaxKey]); }
    *n_out = (uint32_t)keys.size();
    if (keys.size() > cap) return fail(MHX_E_CAPACITY, "export: %zu entries, buffer holds %u", keys.size(), cap);
    if (!keys.empty()) {
        if (!hashes || !counts) return fail(MHX_E_ARG, "null output buffer");
        sort_pairs(keys.data(), cnts.data(), keys.size(), sk->sorted);
        memcpy(hashes, sk->sorted.keys.data(), keys.size() * sizeof(uint64_t));
        memcpy(counts, sk->sorted.cnts.data(), cnts.size() * sizeof(uint32_t));
    }
    return MHX_OK;
}

// Multi-GPU fast path: the shard's partial result as ONE device-resident slab of int64 words
//   [0] n entries (may exceed cap: then only cap are present)   [1] admission threshold T
//   [2] device flags   [3 .. 3+cap) hashes   [3+cap ..) counts, two u32 per word
// holding every (hash, count) with hash <= T (T read on the device), unsorted.  Everything is enqueued on the engine stream
// and the stream is synchronised once, so the slab can go straight into an all-gather; nothing
// crosses PCIe here.
extern "C" int mhx_sketcher_export_slab(mhx_sketcher *sk, void *d_slab, uint32_t cap)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !d_slab || cap == 0 || (cap & 1)) return fail(MHX_E_ARG, "export_slab: null argument or odd capacity");
    rc = settle(sk);
    if (rc) return rc;
    uint64_t *w = (uint64_t *)d_slab;
    HIPCHK(hipMemsetAsync(w, 0, 3 * sizeof(uint64_t), g.stream));
    // (the extract kernel ORs the device flags and the state bits of the m > 1 phase, MHX_SLAB_*, into word [2])
    HIPCHK(launch_extract(table_args(sk), 0, 1, w + 3, (uint32_t *)(w + 3 + cap), cap, (uint32_t *)w, w + 2, sk->d_thresh, w + 1, nullptr, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return MHX_OK;
}

// ---- sharded path: sizes first, slabs sized from the data, merge on the device ------------------
// 1. mhx_sketcher_export_begin : every (hash, count) <= T_r of this shard -> the sketcher's own device buffer; the
//                                 header [n_r, T_r, flags, #(2^64-1), occupied slots] comes back (ranks all-gather it)
// 2. mhx_sketcher_export_pack  : the entries as ONE slab [hashes[cap] | counts u32[cap]], cap = max_r n_r, into the
//                                 caller's send buffer (device memory for RCCL, host memory for gloo)
// 3. mhx_sketcher_merge_slabs  : the other ranks' gathered slabs are added to this rank's candidate table
//                                 (slab_insert_kernel), the ordinary extraction with limit T_min = min_r T_r yields the
//                                 union's sketch; same exactness rule as finish() -> MHX_E_CAPACITY, never a short sketch
static int export_begin_impl(mhx_sketcher *sk, uint64_t *header8)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !header8) return fail(MHX_E_ARG, "null argument");
    if (sk->merged) return fail(MHX_E_ARG, "this sketcher holds a merged table: mhx_sketcher_reset() first");
    rc = settle(sk);
    if (rc) return rc;
    uint64_t *d = sk->d_exp_hdr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // one kernel: entries to d_out_keys / d_out_cnts, the five header words accumulated on the device and stored
        // into the pinned mirror (and cleared for the next call) by the workgroup that finishes last
        HIPCHK(launch_extract(table_args(sk), 0, 1, sk->d_out_keys, sk->d_out_cnts, sk->out_cap, (uint32_t *)d, d + 2, sk->d_thresh, d + 1, d + 3,
                              g.stream, nullptr, 0, d, sk->h_exp_hdr, sk->d_done, d + 4, 5));
        HIPCHK(hipStreamSynchronize(g.stream));
        const uint64_t n = sk->h_exp_hdr[0];
        if (n > sk->out_cap) { // grow (with room for the next, similar shard) and repeat once
            hipFree(sk->d_out_keys); hipFree(sk->d_out_cnts);
            sk->d_out_keys = nullptr; sk->d_out_cnts = nullptr;
            if (n + n / 4 + 1024 > 0xFFFFFFF0ull) return fail(MHX_E_CAPACITY, "export: %llu entries", (unsigned long long)n);
            sk->out_cap = (uint32_t)(n + n / 4 + 1024);
            HIPCHK(hipMalloc((void **)&sk->d_out_keys, (size_t)sk->out_cap * sizeof(uint64_t)));
            HIPCHK(hipMalloc((void **)&sk->d_out_cnts, (size_t)sk->out_cap * sizeof(uint32_t)));
            continue;
        }
        for (int i = 0; i < 5; ++i) header8[i] = sk->h_exp_hdr[i];
        header8[5] = header8[6] = header8[7] = 0;
        sk->exported = n;
        sk->export_valid = true;
        sk->last_T = header8[1];
        return MHX_OK;
    }
    return fail(MHX_E_INTERNAL, "export: output kept growing");
}

extern "C" int mhx_sketcher_export_begin(mhx_sketcher *sk, uint64_t *header8)
{
    try {
        return export_begin_impl(sk, header8);
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_sketcher_export_begin: %s", e.what());
    }
}

extern "C" int mhx_sketcher_export_pack(mhx_sketcher *sk, void *dst, uint64_t cap_entries)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !dst) return fail(MHX_E_ARG, "null argument");
    if (!sk->export_valid) return fail(MHX_E_ARG, "export_pack without a preceding mhx_sketcher_export_begin");
    if ((cap_entries & 1) || cap_entries < sk->exported) return fail(MHX_E_ARG, "export_pack: capacity %llu is odd or below this shard's %llu entries",
                                                                     (unsigned long long)cap_entries, (unsigned long long)sk->exported);
    uint8_t *p = (uint8_t *)dst;
    if (sk->exported) { // device-to-device for RCCL send buffers, device-to-host for gloo's
        HIPCHK(hipMemcpyAsync(p, sk->d_out_keys, sk->exported * sizeof(uint64_t), hipMemcpyDefault, g.stream));
        HIPCHK(hipMemcpyAsync(p + cap_entries * sizeof(uint64_t), sk->d_out_cnts, sk->exported * sizeof(uint32_t), hipMemcpyDefault, g.stream));
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    return MHX_OK;
}

static int merge_slabs_impl(mhx_sketcher *sk, const void *slabs, int slabs_on_device, uint32_t n_ranks, uint64_t cap_entries,
                            const uint64_t *headers, uint32_t own_rank, uint64_t *hashes, uint32_t *counts, uint32_t *n_out,
                            uint32_t hdr_words = 0)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !headers || !hashes || !n_out || n_ranks == 0) return fail(MHX_E_ARG, "null argument");
    if (own_rank >= n_ranks) return fail(MHX_E_ARG, "own_rank %u out of range (%u ranks)", own_rank, n_ranks);
    if (cap_entries & 1) return fail(MHX_E_ARG, "merge_slabs: odd slab capacity");
    if (sk->merged) return fail(MHX_E_ARG, "this sketcher holds a merged table already: mhx_sketcher_reset() first");
    if (!sk->export_valid) return fail(MHX_E_ARG, "merge_slabs without a preceding mhx_sketcher_export_begin on this sketcher");
    uint64_t t_min = ~0ull, others = 0, maxkey_others = 0, flags = 0, max_n = 0;
    for (uint32_t r = 0; r < n_ranks; ++r) {
        const uint64_t *h = headers + 8 * (size_t)r;
        if (h[0] > cap_entries) return fail(MHX_E_ARG, "merge_slabs: rank %u announces %llu entries, slabs hold %llu", r, (unsigned long long)h[0], (unsigned long long)cap_entries);
        t_min = h[1] < t_min ? h[1] : t_min;
        flags |= h[2] & kFlagErrorMask & ~kFlagNeedLookback;
        if (r != own_rank) { others += h[0]; maxkey_others += h[3]; max_n = h[0] > max_n ? h[0] : max_n; }
    }
    if (headers[8 * (size_t)own_rank] != sk->exported || headers[8 * (size_t)own_rank + 1] != sk->last_T)
        return fail(MHX_E_ARG, "merge_slabs: header of rank %u is not this sketcher's export", own_rank);
    rc = check_flags(flags); // a full table or a malformed FASTQ on ANY rank
    if (rc) return rc;
    if ((others || n_ranks > 1) && !slabs) return fail(MHX_E_ARG, "null slabs");
    const uint64_t slab_words = hdr_words + cap_entries + cap_entries / 2;
    const uint64_t *d_slabs = (const uint64_t *)slabs;
    if (!slabs_on_device && (others || headers[8 * (size_t)own_rank])) {
        const size_t bytes = (size_t)n_ranks * slab_words * sizeof(uint64_t);
        if (sk->merge_in_cap < bytes) {
            HIPCHK(hipStreamSynchronize(g.stream));
            hipFree(sk->d_merge_in);
            sk->d_merge_in = nullptr;
            sk->merge_in_cap = 0;
            const size_t cap = (bytes + bytes / 4 + (1u << 20)) & ~(size_t)((1u << 20) - 1);
            HIPCHK(hipMalloc((void **)&sk->d_merge_in, cap));
            sk->merge_in_cap = cap;
        }
        HIPCHK(hipMemcpyAsync(sk->d_merge_in, slabs, bytes, hipMemcpyHostToDevice, g.stream));
        d_slabs = sk->d_merge_in;
    }
    sk->merged = true;
    // The usual case: all slabs (this rank's own among them) are binned by value and merged bin by bin in LDS; the result
    // lands in the pinned block in hash order.  Non-uniform data (a bin overflows), more than 64 ranks or more than ~16 M
    // entries take the table path below.
    static const bool force_table = getenv("MHX_MERGE_TABLE") != nullptr;
    const uint64_t total = others + headers[8 * (size_t)own_rank];
    if (!force_table && n_ranks <= kMaxMergeRanks && total > 0 && total <= (uint64_t)kMergeMaxBins * 1024) {
        uint32_t nbins = 256;
        while ((uint64_t)nbins * 1024 < total) nbins <<= 1;
        const uint32_t lg = (uint32_t)__builtin_ctz(nbins);
        const uint32_t bits = 64u - (uint32_t)__builtin_clzll(t_min | 1ull);
        MergeArgs a;
        a.shift = bits > lg ? bits - lg : 0u;
        const uint64_t bins_used = (t_min >> a.shift) + 1;
        const double avg = (double)total / (double)bins_used;
        a.region = (uint32_t)(avg + 6.0 * sqrt(avg) + 64.0);
        a.table_slots = 256;
        while ((uint64_t)a.table_slots * 3 / 4 < a.region) a.table_slots <<= 1;
        if (a.table_slots <= kMergeMaxSlots) {
            if (!sk->d_mg_small) {
                HIPCHK(hipMalloc((void **)&sk->d_mg_small, (2 * (size_t)kMergeMaxBins + 16) * sizeof(uint32_t)));
                HIPCHK(hipMemsetAsync(sk->d_mg_small, 0, (2 * (size_t)kMergeMaxBins + 16) * sizeof(uint32_t), g.stream));
            }
            const size_t need = (size_t)nbins * a.region;
            if (sk->mg_entries < need) {
                HIPCHK(hipStreamSynchronize(g.stream));
                hipFree(sk->d_mg_keys); hipFree(sk->d_mg_cnts);
                sk->d_mg_keys = nullptr; sk->d_mg_cnts = nullptr;
                sk->mg_entries = 0;
                const size_t want = need + need / 4;
                HIPCHK(hipMalloc((void **)&sk->d_mg_keys, want * sizeof(uint64_t)));
                HIPCHK(hipMalloc((void **)&sk->d_mg_cnts, want * sizeof(uint32_t)));
                sk->mg_entries = want;
            }
            a.slabs = d_slabs; a.slab_words = slab_words; a.cap = cap_entries; a.hdr_words = hdr_words; a.nranks = n_ranks; a.min_mult = sk->m; a.t_min = t_min; a.nbins = nbins;
            uint64_t max_all = 0;
            for (uint32_t r = 0; r < kMaxMergeRanks; ++r) { a.n[r] = r < n_ranks ? headers[8 * (size_t)r] : 0; max_all = a.n[r] > max_all ? a.n[r] : max_all; }
            a.cursor = sk->d_mg_small; a.qn = sk->d_mg_small + kMergeMaxBins; a.flags = sk->d_mg_small + 2 * kMergeMaxBins;
            a.sc_keys = sk->d_mg_keys; a.sc_cnts = sk->d_mg_cnts;
            HIPCHK(launch_merge_bins(a, max_all, sk->h_fin, sk->fin_cap, g.stream));
            HIPCHK(hipStreamSynchronize(g.stream));
            const uint64_t *h = sk->h_fin;
            const uint64_t n_q = h[0];
            static const bool dbg = getenv("MHX_MERGE_DEBUG") != nullptr;
            if (dbg) fprintf(stderr, "[mhx merge] %u ranks, %llu entries, %u bins (%llu used) of %u entries, table %u: %llu qualify, flags %llu\n", n_ranks,
                             (unsigned long long)total, nbins, (unsigned long long)bins_used, a.region, a.table_slots, (unsigned long long)n_q, (unsigned long long)h[2]);
            if (h[2] == 0) { // (more qualify than the block holds? the bins are in value order: its first s entries are the sketch)
                const uint64_t maxkey_all = maxkey_others + headers[8 * (size_t)own_rank + 3];
                const bool extra = t_min == ~0ull && maxkey_all >= sk->m; // the one hash value no table holds
                const uint64_t n_src = n_q + (extra ? 1 : 0);
                if (n_src < sk->s && t_min < sk->hash_max)
                    return fail(MHX_E_CAPACITY, "sharded sketch not exact: %llu of %u entries with multiplicity >= %u below the smallest shard threshold; "
                                "every rank must sketch its shard again with a larger budget_scale", (unsigned long long)n_src, sk->s, sk->m);
                const uint32_t nn = n_src < sk->s ? (uint32_t)n_src : sk->s;
                const uint32_t from_block = nn < n_q ? nn : (uint32_t)n_q; // <= s <= fin_cap: all of them are in the block
                memcpy(hashes, h + 4, (size_t)from_block * sizeof(uint64_t));
                if (counts) memcpy(counts, reinterpret_cast<const uint32_t *>(h + 4 + sk->fin_cap), (size_t)from_block * sizeof(uint32_t));
                if (nn > from_block) { hashes[from_block] = ~0ull; if (counts) counts[from_block] = maxkey_all > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)maxkey_all; }
                *n_out = nn;
                return MHX_OK;
            }
            // (flags raised -- a bin's region or table overflowed on non-uniform data: the table path decides)
        }
    }
    const uint64_t occupied = headers[8 * (size_t)own_rank + 4];
    if (occupied + others > sk->nslots / 2) {
        // The other shards' entries would crowd this table (tiny tables of tiny inputs, or shards that never tightened
        // their thresholds): the host merge decides instead, by the same rule on the same gathered data.
        std::vector<uint64_t> hbuf((size_t)n_ranks * slab_words);
        if (slabs_on_device) {
            HIPCHK(hipMemcpyAsync(hbuf.data(), slabs, hbuf.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipStreamSynchronize(g.stream));
        } else {
            memcpy(hbuf.data(), slabs, hbuf.size() * sizeof(uint64_t));
        }
        std::vector<uint64_t> ah, an(n_ranks), at(n_ranks);
        std::vector<uint32_t> ac;
        for (uint32_t r = 0; r < n_ranks; ++r) {
            const uint64_t n = headers[8 * (size_t)r], mk = headers[8 * (size_t)r + 3];
            const uint64_t *hp = hbuf.data() + (size_t)r * slab_words + hdr_words;
            const uint32_t *cp = reinterpret_cast<const uint32_t *>(hp + cap_entries);
            ah.insert(ah.end(), hp, hp + n);
            ac.insert(ac.end(), cp, cp + n);
            an[r] = n;
            at[r] = headers[8 * (size_t)r + 1];
            if (mk && at[r] == ~0ull) { ah.push_back(~0ull); ac.push_back(mk > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)mk); ++an[r]; } // the one value the table cannot hold
        }
        sk->merged = true;
        return mhx_merge_shard_partials(ah.data(), ac.data(), an.data(), at.data(), n_ranks, sk->k, sk->s, sk->m, hashes, counts, n_out);
    }
    for (uint32_t r0 = 0; r0 < n_ranks; r0 += kMaxMergeRanks) { // (one launch for up to 64 ranks)
        SlabMergeArgs a;
        a.slabs = d_slabs + (size_t)r0 * slab_words;
        a.slab_words = slab_words;
        a.cap = cap_entries;
        a.hdr_words = hdr_words;
        a.nranks = n_ranks - r0 < kMaxMergeRanks ? n_ranks - r0 : kMaxMergeRanks;
        for (uint32_t r = 0; r < kMaxMergeRanks; ++r) a.n[r] = r < a.nranks ? headers[8 * (size_t)(r0 + r)] : 0;
        a.own_rank = own_rank >= r0 && own_rank - r0 < a.nranks ? own_rank - r0 : kMaxMergeRanks;
        a.t_min = t_min;
        a.maxkey_others = r0 == 0 ? maxkey_others : 0;
        a.keys = sk->d_keys; a.cnts = sk->d_cnts; a.slot_mask = sk->nslots - 1; a.thresh = sk->d_thresh; a.stats = sk->d_stats;
        HIPCHK(launch_slab_insert(a, max_n, g.stream));
    }
    // the table now holds the union below T_min with summed counts, and T = T_min on the device: the ordinary extraction
    // (count >= m, hash <= T, ordering kernels for large sketches) and finish()'s exactness rule do the rest
    sk->table_dirty = false;
    sk->table_sampled = false;
    sk->unsettled.clear();
    return mhx_sketcher_finish(sk, hashes, counts, n_out);
}

extern "C" int mhx_sketcher_merge_slabs(mhx_sketcher *sk, const void *slabs, int slabs_on_device, uint32_t n_ranks, uint64_t cap_entries,
                                        const uint64_t *headers, uint32_t own_rank, uint64_t *hashes, uint32_t *counts, uint32_t *n_out)
{
    try {
        return merge_slabs_impl(sk, slabs, slabs_on_device, n_ranks, cap_entries, headers, own_rank, hashes, counts, n_out);
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_sketcher_merge_slabs: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_sketcher_merge_slabs: %s", e.what());
    }
}

// ---- the same exchange in ONE collective when the slabs live on the device (RCCL) -----------------------------------
// The 64-byte header rides in front of the slab: [header8 | hashes[cap] | counts u32[cap]], cap = the caller's guess (the
// last exchange's sizes, or 4 s + 4096 the first time).  mhx_sketcher_export_into writes the shard's partial result
// straight into the caller's send buffer -- no separate compaction buffer, no pack step --, the ranks all-gather the
// slabs, and mhx_sketcher_merge_gathered reads the gathered headers back itself: sizes first is then "sizes with", and
// only when some rank holds more entries than the guess does the caller repeat with the capacity that call reports
// (*need_cap; every rank sees the same headers and takes the same turn).
extern "C" int mhx_sketcher_export_into(mhx_sketcher *sk, void *d_slab, uint64_t cap_entries, uint64_t *header8)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !d_slab || !header8 || cap_entries == 0 || (cap_entries & 1) || cap_entries > 0xFFFFFFF0ull) return fail(MHX_E_ARG, "export_into: null argument or bad capacity");
    if (sk->merged) return fail(MHX_E_ARG, "this sketcher holds a merged table: mhx_sketcher_reset() first");
    rc = settle(sk);
    if (rc) return rc;
    uint64_t *w = (uint64_t *)d_slab, *d = sk->d_exp_hdr;
    HIPCHK(launch_extract(table_args(sk), 0, 1, w + 8, (uint32_t *)(w + 8 + cap_entries), (uint32_t)cap_entries, (uint32_t *)d, d + 2, sk->d_thresh, d + 1, d + 3,
                          g.stream, nullptr, 0, d, sk->h_exp_hdr, sk->d_done, d + 4, 8, w));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int i = 0; i < 8; ++i) header8[i] = sk->h_exp_hdr[i];
    sk->exported = header8[0];
    sk->export_valid = true;
    sk->last_T = header8[1];
    return MHX_OK;
}

extern "C" int mhx_sketcher_merge_gathered(mhx_sketcher *sk, const void *d_slabs, uint32_t n_ranks, uint64_t cap_entries, uint32_t own_rank,
                                           uint64_t *hashes, uint32_t *counts, uint32_t *n_out, uint64_t *need_cap)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !d_slabs || !hashes || !n_out || !need_cap || n_ranks == 0 || (cap_entries & 1)) return fail(MHX_E_ARG, "merge_gathered: null argument or odd capacity");
    *need_cap = 0;
    try {
        const uint64_t slab_words = 8 + cap_entries + cap_entries / 2;
        std::vector<uint64_t> headers((size_t)n_ranks * 8);
        // the gathered headers: 64 bytes at the front of every slab
        HIPCHK(hipMemcpy2DAsync(headers.data(), 64, d_slabs, slab_words * sizeof(uint64_t), 64, n_ranks, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        uint64_t max_n = 0;
        for (uint32_t r = 0; r < n_ranks; ++r) max_n = std::max(max_n, headers[8 * (size_t)r]);
        if (max_n > cap_entries) { // some slab is cut short: the caller repeats the exchange with room for all of it
            *need_cap = max_n;
            return fail(MHX_E_CAPACITY, "merge_gathered: a shard holds %llu entries, the slabs %llu", (unsigned long long)max_n, (unsigned long long)cap_entries);
        }
        return merge_slabs_impl(sk, d_slabs, 1, n_ranks, cap_entries, headers.data(), own_rank, hashes, counts, n_out, 8);
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_sketcher_merge_gathered: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_sketcher_merge_gathered: %s", e.what());
    }
}

// Union of shard partials: sum the counts of equal hashes, keep count >= m, first s.
extern "C" int mhx_merge_partials(const uint64_t *hashes, const uint32_t *counts, uint64_t n, uint32_t s, uint32_t min_mult,
                                  uint64_t *out_hashes, uint32_t *out_counts, uint32_t *n_out)
{
    clear_error();
    if ((!hashes || !counts) && n) return fail(MHX_E_ARG, "null input");
    if (!out_hashes || !n_out) return fail(MHX_E_ARG, "null output");
    std::vector<uint64_t> idx(n);
    for (uint64_t i = 0; i < n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return hashes[a] < hashes[b]; });
    uint32_t w = 0;
    const uint32_t m = min_mult ? min_mult : 1;
    for (uint64_t i = 0; i < n && w < s;) {
        uint64_t j = i, c = 0;
        while (j < n && hashes[idx[j]] == hashes[idx[i]]) c += counts[idx[j++]];
        if (c >= m) {
            out_hashes[w] = hashes[idx[i]];
            if (out_counts) out_counts[w] = c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c;
            ++w;
        }
        i = j;
    }
    *n_out = w;
    return MHX_OK;
}

// The merge step of the sharded path WITH its exactness rule (what finish() checks on one GPU, applied to
// the union).  Below T_min = min_r T_r every shard's list is complete and its counts exact (a shard's
// threshold only ever falls, so a hash <= its final T_r was admitted on every occurrence).  Hence:
//   >= s merged entries with summed count >= m lie <= T_min  -> the first s are the sketch of the union;
//   T_min == hash_max (no shard ever rejected anything)      -> whatever qualifies is the (short) sketch;
//   otherwise the bound was too tight for this input         -> MHX_E_CAPACITY, never a short sketch.
// The decision uses gathered data only, so every rank reaches the same verdict.
extern "C" int mhx_merge_shard_partials(const uint64_t *hashes, const uint32_t *counts, const uint64_t *shard_n,
                                        const uint64_t *shard_threshold, uint32_t n_shards, int k, uint32_t s, uint32_t min_mult,
                                        uint64_t *out_hashes, uint32_t *out_counts, uint32_t *n_out)
{
    clear_error();
    if (!shard_n || !shard_threshold || n_shards == 0) return fail(MHX_E_ARG, "null shard description");
    if (!out_hashes || !n_out || s == 0) return fail(MHX_E_ARG, "null output");
    if (k < 1 || k > 32) return fail(MHX_E_ARG, "k-mer size %d not supported (1..32)", k);
    const uint64_t hash_max = k <= 16 ? 0xFFFFFFFFull : ~0ull;
    uint64_t t_min = ~0ull, total = 0;
    for (uint32_t r = 0; r < n_shards; ++r) {
        t_min = shard_threshold[r] < t_min ? shard_threshold[r] : t_min;
        total += shard_n[r];
    }
    if ((!hashes || !counts) && total) return fail(MHX_E_ARG, "null input");
    try {
        std::vector<uint64_t> h;
        std::vector<uint32_t> c;
        h.reserve(total);
        c.reserve(total);
        for (uint64_t i = 0; i < total; ++i)
            if (hashes[i] <= t_min) { h.push_back(hashes[i]); c.push_back(counts[i]); }
        int rc = mhx_merge_partials(h.data(), c.data(), h.size(), s, min_mult, out_hashes, out_counts, n_out);
        if (rc) return rc;
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_merge_shard_partials: out of host memory");
    }
    if (*n_out < s && t_min < hash_max)
        return fail(MHX_E_CAPACITY, "sharded sketch not exact: %u of %u entries with multiplicity >= %u below the smallest shard threshold; "
                    "every rank must sketch its shard again with a larger budget_scale", *n_out, s, min_mult ? min_mult : 1);
    return MHX_OK;
}

// ---- batched distance ------------------------------------------------------------------
extern "C" double mhx_last_dist_kernel_ms(void) { return g.last_dist_ms; }
extern "C" int mhx_last_dist_fallback_blocks(void) { return g.last_dist_fallbacks; }

// Persistent device staging of the host-pointer form (one buffer, grown on demand): six hipMalloc / hipFree pairs per
// call cost more than the kernels of an AuriClass-sized comparison (1 query x 24 references).
static int dist_stage(size_t bytes, uint8_t **out)
{
    if (g.dist_in_cap < bytes) {
        HIPCHK(hipStreamSynchronize(g.stream));
        hipFree(g.dist_in);
        g.dist_in = nullptr;
        g.dist_in_cap = 0;
        const size_t cap = (bytes + bytes / 4 + (1u << 20)) & ~(size_t)((1u << 20) - 1);
        if (hipMalloc((void **)&g.dist_in, cap) != hipSuccess) return fail(MHX_E_HIP, "hipMalloc failed in dist_batch (%zu bytes)", cap);
        g.dist_in_cap = cap;
    }
    *out = g.dist_in;
    return MHX_OK;
}

// q_rows / r_rows (host form only): the rows where they lie, one pointer each (q / r are then unused) -- mhx_dist_files
// hands over the hash lists inside its pinned image of the reference sketch file instead of building padded matrices
static int dist_batch_core(const uint64_t *q, const uint32_t *q_len, uint32_t nq, const uint64_t *r, const uint32_t *r_len,
                           uint32_t nr, uint32_t stride, int k, uint32_t s, uint32_t *common, uint32_t *denom, double *dist,
                           int device_ptrs, const uint64_t *const *q_rows, const uint64_t *const *r_rows)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (nq == 0 || nr == 0) return MHX_OK;
    if ((!q && !q_rows) || !q_len || (!r && !r_rows) || !r_len || !common || !denom) return fail(MHX_E_ARG, "null argument");
    if (device_ptrs && (q_rows || r_rows)) return fail(MHX_E_ARG, "row pointers are a host form");
    if (k < 1 || k > 32 || s == 0 || stride == 0) return fail(MHX_E_ARG, "bad k / s / stride");
    const uint64_t pairs = (uint64_t)nq * nr;
    if (pairs > 0x7FFFFFFFull) return fail(MHX_E_ARG, "too many pairs for one call");
    DistArgs a;
    a.nq = nq; a.nr = nr; a.stride = stride; a.s = s; a.k = k; a.out_stride = nr; a.out_off = 0;
    uint32_t *dc = nullptr, *dd = nullptr;
    if (device_ptrs) {
        a.q = q; a.q_len = q_len; a.r = r; a.r_len = r_len; a.common = common; a.denom = denom; a.dist = dist;
    } else {
        for (uint32_t i = 0; i < nq; ++i) if (q_len[i] > stride) return fail(MHX_E_ARG, "q_len[%u] exceeds stride", i);
        for (uint32_t i = 0; i < nr; ++i) if (r_len[i] > stride) return fail(MHX_E_ARG, "r_len[%u] exceeds stride", i);
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t bq = up((size_t)nq * stride * 8), br = up((size_t)nr * stride * 8), bql = up((size_t)nq * 4), brl = up((size_t)nr * 4), bo = up(pairs * 4);
        uint8_t *base = nullptr;
        rc = dist_stage(bq + br + bql + brl + 2 * bo, &base);
        if (rc) return rc;
        uint8_t *dq = base, *dr = dq + bq, *dql = dr + br, *drl = dql + bql;
        dc = (uint32_t *)(drl + brl);
        dd = (uint32_t *)(drl + brl + bo);
        // rows that are mostly padding travel one by one (valid prefix only), full ones as one block
        hipError_t ce = hipSuccess;
        auto rows = [&](uint8_t *dst, const uint64_t *src, const uint32_t *len, uint32_t n, const uint64_t *const *ptrs) {
            if (ptrs) { // every row from its own place
                for (uint32_t i = 0; i < n && ce == hipSuccess; ++i)
                    if (len[i]) ce = hipMemcpyAsync(dst + (size_t)i * stride * 8, ptrs[i], (size_t)len[i] * 8, hipMemcpyHostToDevice, g.stream);
                return;
            }
            uint64_t valid = 0;
            for (uint32_t i = 0; i < n; ++i) valid += len[i];
            if (n > 64 || valid * 2 >= (uint64_t)n * stride) {
                if (ce == hipSuccess) ce = hipMemcpyAsync(dst, src, (size_t)n * stride * 8, hipMemcpyHostToDevice, g.stream);
                return;
            }
            for (uint32_t i = 0; i < n && ce == hipSuccess; ++i)
                if (len[i]) ce = hipMemcpyAsync(dst + (size_t)i * stride * 8, src + (size_t)i * stride, (size_t)len[i] * 8, hipMemcpyHostToDevice, g.stream);
        };
        rows(dq, q, q_len, nq, q_rows);
        rows(dr, r, r_len, nr, r_rows);
        if (ce == hipSuccess) ce = hipMemcpyAsync(dql, q_len, (size_t)nq * 4, hipMemcpyHostToDevice, g.stream);
        if (ce == hipSuccess) ce = hipMemcpyAsync(drl, r_len, (size_t)nr * 4, hipMemcpyHostToDevice, g.stream);
        if (ce != hipSuccess) return fail(MHX_E_HIP, "H2D copy failed in dist_batch: %s", hipGetErrorString(ce));
        a.q = (const uint64_t *)dq; a.q_len = (const uint32_t *)dql; a.r = (const uint64_t *)dr; a.r_len = (const uint32_t *)drl;
        a.common = dc; a.denom = dd; a.dist = nullptr; // distances in host libm below
    }
    // all-vs-refs fast path: the references go through in slices of 32 (one bit each in the range kernel's masks), the
    // queries in batches (MHX_DIST_QBATCH; default: all at once), every (batch, slice) filling its block of the [nq][nr]
    // outputs; the generic pair-per-workgroup kernel serves tiny batches and is the fallback of a block whose value
    // ranges are too uneven for the LDS table.  Nothing is read back between the blocks: every block has its own flag
    // word, all of them come back with ONE copy behind the last launch.
    // (few pairs of LONG lists take it too -- AuriClass's own call, 1 query x 24 references at s = 50 000: 0.48 ms in the
    // generic kernel, whose 24 workgroups each walk 100 000 elements)
    const bool fast = (pairs >= 64 || (pairs >= 8 && pairs * (uint64_t)s >= 400000)) && getenv("MHX_DIST_GENERIC") == nullptr;
    uint32_t qbatch = nq;
    if (const char *e = getenv("MHX_DIST_QBATCH")) { const long v = atol(e); if (v > 0 && (uint64_t)v < nq) qbatch = (uint32_t)v; }
    const uint32_t nslices = (nr + 31) / 32, nbatches = (nq + qbatch - 1) / qbatch, nblocks = nslices * nbatches;
    DistWork w{};
    uint32_t *d_params = nullptr;
    constexpr uint32_t kBlockGroup = 4096; // blocks whose flag words come back together (a reference set of 131 072 sketches per group)
    if (fast) {
        size_t oq, orr, oc, op;
        const size_t need = dist_work_bytes(qbatch, nr < 32 ? nr : 32, &oq, &orr, &oc, &op) + (size_t)std::min(nblocks, kBlockGroup) * 8;
        if (g.dist_ws_cap < need) {
            HIPCHK(hipStreamSynchronize(g.stream));
            hipFree(g.dist_ws);
            g.dist_ws = nullptr;
            g.dist_ws_cap = 0;
            if (hipMalloc((void **)&g.dist_ws, need) != hipSuccess) return fail(MHX_E_HIP, "hipMalloc failed for the distance workspace");
            g.dist_ws_cap = need;
        }
        w.offs_q = (uint32_t *)(g.dist_ws + oq); w.offs_r = (uint32_t *)(g.dist_ws + orr);
        w.cpart = g.dist_ws + oc;
        d_params = (uint32_t *)(g.dist_ws + op); // [block][2]: shift, overflow flag
    }
    auto block_args = [&](uint32_t b) {
        const uint32_t q0 = (b / nslices) * qbatch, r0 = (b % nslices) * 32;
        DistArgs x = a;
        x.q = a.q + (uint64_t)q0 * stride;
        x.q_len = a.q_len + q0;
        x.nq = nq - q0 < qbatch ? nq - q0 : qbatch;
        x.r = a.r + (uint64_t)r0 * stride;
        x.r_len = a.r_len + r0;
        x.nr = nr - r0 < 32 ? nr - r0 : 32;
        x.common = a.common + (uint64_t)q0 * nr;
        x.denom = a.denom + (uint64_t)q0 * nr;
        x.dist = a.dist ? a.dist + (uint64_t)q0 * nr : nullptr;
        x.out_off = r0;
        return x;
    };
    hipEventRecord(g.ev0, g.stream);
    hipError_t le = hipSuccess;
    if (!fast) { le = launch_dist_pairs(a, g.stream); g.last_dist_fallbacks = -1; }
    if (fast) g.last_dist_fallbacks = 0;
    for (uint32_t b0 = 0; fast && b0 < nblocks && le == hipSuccess; b0 += kBlockGroup) {
        const uint32_t b1 = std::min(nblocks, b0 + kBlockGroup);
        for (uint32_t b = b0; b < b1 && le == hipSuccess; ++b) {
            w.params = d_params + 2 * (b - b0);
            le = launch_dist_ranges(block_args(b), w, g.stream);
        }
        if (le != hipSuccess) break;
        std::vector<uint32_t> flags((size_t)(b1 - b0) * 2);
        if (hipMemcpyAsync(flags.data(), d_params, flags.size() * 4, hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
            hipStreamSynchronize(g.stream) != hipSuccess)
            return fail(MHX_E_HIP, "dist kernel failed");
        for (uint32_t b = b0; b < b1 && le == hipSuccess; ++b)
            if (flags[2 * (b - b0) + 1]) { le = launch_dist_pairs(block_args(b), g.stream); ++g.last_dist_fallbacks; } // a value range overflowed the LDS table
    }
    hipEventRecord(g.ev1, g.stream);
    if (le != hipSuccess) return fail(MHX_E_HIP, "dist kernel launch failed: %s", hipGetErrorString(le));
    hipError_t se = hipSuccess;
    if (!device_ptrs) {
        se = hipMemcpyAsync(common, dc, pairs * 4, hipMemcpyDeviceToHost, g.stream);
        if (se == hipSuccess) se = hipMemcpyAsync(denom, dd, pairs * 4, hipMemcpyDeviceToHost, g.stream);
    }
    if (se == hipSuccess) se = hipStreamSynchronize(g.stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, g.ev0, g.ev1);
    g.last_dist_ms = ms;
    if (se != hipSuccess) return fail(MHX_E_HIP, "dist kernel failed: %s", hipGetErrorString(se));
    if (!device_ptrs && dist) {
        for (uint64_t i = 0; i < pairs; ++i) {
            double d;
            if (common[i] == denom[i]) d = 0.0;
            else if (common[i] == 0) d = 1.0;
            else {
                const double j = (double)common[i] / (double)denom[i];
                d = -log(2.0 * j / (1.0 + j)) / (double)k;
                if (d > 1.0) d = 1.0;
            }
            dist[i] = d;
        }
    }
    return MHX_OK;
}

extern "C" int mhx_dist_batch(const uint64_t *q, const uint32_t *q_len, uint32_t nq, const uint64_t *r, const uint32_t *r_len,
                              uint32_t nr, uint32_t stride, int k, uint32_t s, uint32_t *common, uint32_t *denom, double *dist,
                              int device_ptrs)
{
    try {
        return dist_batch_core(q, q_len, nq, r, r_len, nr, stride, k, s, common, denom, dist, device_ptrs, nullptr, nullptr);
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_dist_batch: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_dist_batch: %s", e.what());
    }
}

namespace mhx {
int dist_batch_rows(const uint64_t *const *q_rows, const uint32_t *q_len, uint32_t nq, const uint64_t *const *r_rows, const uint32_t *r_len,
                    uint32_t nr, int k, uint32_t s, uint32_t *common, uint32_t *denom, double *dist)
{
    uint32_t stride = 16;
    for (uint32_t i = 0; i < nq; ++i) stride = q_len[i] > stride ? q_len[i] : stride;
    for (uint32_t i = 0; i < nr; ++i) stride = r_len[i] > stride ? r_len[i] : stride;
    stride = (stride + 15u) & ~15u; // rows of whole 128-byte lines on the device
    return dist_batch_core(nullptr, q_len, nq, nullptr, r_len, nr, stride, k, s, common, denom, dist, 0, q_rows, r_rows);
}
} // namespace mhx

extern "C" int mhx_sketcher_finish(mhx_sketcher *sk, uint64_t *hashes, uint32_t *counts, uint32_t *n_out)
{
    try {
        return mhx_sketcher_finish_impl(sk, hashes, counts, n_out);
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_sketcher_finish: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_sketcher_finish: %s", e.what());
    }
}

extern "C" int mhx_sketcher_export(mhx_sketcher *sk, uint64_t limit, uint64_t *hashes, uint32_t *counts, uint32_t cap, uint32_t *n_out)
{
    try {
        return mhx_sketcher_export_impl(sk, limit, hashes, counts, cap, n_out);
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_sketcher_export: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_sketcher_export: %s", e.what());
    }
}
