// mhx_engine.cpp -- host side of libmhx: device selection, the sketcher object that
// schedules tile launches and threshold tightening, the batched distance entry point and
// the two file-level calls that replace AuriClass's `mash sketch` / `mash dist`
// subprocesses (/root/reference/auriclass/classes.py:576-596, 696-713, 92-104).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <exception>
#include <new>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mhx_device.h"
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

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(MHX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

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
    // bulk file ingest: pinned staging ring + copy stream (allocated on first use, kept)
    static constexpr int kPinnedSlots = 4;
    uint8_t *pinned[kPinnedSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pinned_free[kPinnedSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
};
static Engine g;

static int require_engine()
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
    hipFree(g.dist_ws);
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
struct mhx_sketcher {
    int k = 0;
    uint32_t s = 0, m = 1;
    bool hash32 = false;
    uint64_t nslots = 0;
    uint64_t hash_max = 0;   // largest representable hash (2^64-1 or 2^32-1)
    uint64_t t_init = 0;     // initial admission threshold (everything admitted)
    uint64_t t_write = 0;    // staging word for thresholds the host imposes
    // device
    uint64_t *d_keys = nullptr;
    uint32_t *d_cnts = nullptr;
    uint64_t *d_thresh = nullptr;
    uint32_t *d_hist = nullptr;
    uint64_t *d_acc = nullptr;
    uint64_t *d_stats = nullptr;   // kStatReplicas x kStatCount
    uint32_t *d_tickets = nullptr; // one per launch of a push
    uint64_t *d_tile_state = nullptr;
    size_t tile_state_cap = 0;
    uint8_t *d_stage = nullptr;
    size_t stage_cap = 0;
    uint64_t *d_out_keys = nullptr;
    uint32_t *d_out_cnts = nullptr;
    uint32_t *d_out_n = nullptr;
    uint32_t out_cap = 0;
    // host
    uint64_t next_chunk_bytes = 0; // geometric schedule of the tightening phase
    bool settled = false;          // threshold tight enough: remaining data goes in one launch
    uint64_t settled_total = 0;    // input size that decision was made for
    uint64_t bytes_pushed = 0;
    uint64_t expected_bytes = 0;
    uint64_t admit_scale = 1;      // multiplies the initial admission budget (retries after MHX_E_CAPACITY)
    double hash_ms = 0.0;
    uint64_t launches = 0;
    uint64_t last_T = 0;
    // m > 1 only: until s hashes with count >= m exist below T the table is protected by a bound that
    // follows the input seen so far (see push_device)
    bool bounded = false;      // a host-imposed bound has limited T at least once
    bool established = false;  // the tighten pass has lowered T from solid (count >= m) entries
    uint64_t occupied = 0;     // table occupancy reported by the last tighten pass
    uint64_t solid = 0;        // entries <= T with count >= m reported by the last tighten pass
};

static constexpr int kMaxLaunchesPerPush = 64;
#ifndef MHX_CHUNK_GROWTH
#define MHX_CHUNK_GROWTH 16
#endif
static constexpr uint64_t kChunkGrowth = MHX_CHUNK_GROWTH; // chunk size ratio between tighten rounds
static constexpr uint64_t kUncappedBytes = 1u << 20;       // m > 1: prefix of the input that is admitted whole

static TableArgs table_args(mhx_sketcher *sk)
{
    TableArgs t;
    t.keys = sk->d_keys; t.cnts = sk->d_cnts; t.nslots = sk->nslots; t.thresh = sk->d_thresh;
    t.hist = sk->d_hist; t.acc = sk->d_acc; t.stats = sk->d_stats; t.min_mult = sk->m; t.sketch_size = sk->s;
    t.sample = 1;
    return t;
}

static void free_sketcher(mhx_sketcher *sk)
{
    if (!sk) return;
    hipFree(sk->d_keys); hipFree(sk->d_cnts); hipFree(sk->d_thresh); hipFree(sk->d_hist); hipFree(sk->d_acc);
    hipFree(sk->d_stats); hipFree(sk->d_tickets); hipFree(sk->d_tile_state); hipFree(sk->d_stage);
    hipFree(sk->d_out_keys); hipFree(sk->d_out_cnts); hipFree(sk->d_out_n);
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
    HIPCHK(hipMemsetAsync(sk->d_keys, 0xFF, sk->nslots * sizeof(uint64_t), g.stream));
    HIPCHK(hipMemsetAsync(sk->d_cnts, 0, sk->nslots * sizeof(uint32_t), g.stream));
    HIPCHK(hipMemsetAsync(sk->d_hist, 0, kHistBins * sizeof(uint32_t), g.stream));
    HIPCHK(hipMemsetAsync(sk->d_acc, 0, kAccReplicas * 8 * sizeof(uint64_t), g.stream));
    HIPCHK(hipMemsetAsync(sk->d_stats, 0, kStatReplicas * kStatCount * sizeof(uint64_t), g.stream));
    // Admission threshold: everything is admitted at first.  For m = 1 the first tighten pass already
    // finds s entries; for m > 1 push_device keeps the table safe until s solid hashes exist.
    sk->t_init = sk->hash_max;
    sk->last_T = sk->hash_max;
    sk->bounded = false;
    sk->established = false;
    sk->occupied = 0;
    sk->solid = 0;
    // t_init lives in the sketcher and is only written at creation, so the copy may stay in flight
    HIPCHK(hipMemcpyAsync(sk->d_thresh, &sk->t_init, sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
    uint64_t c0 = next_pow2((uint64_t)sk->s * 64);
    if (c0 < (256u << 10)) c0 = 256u << 10;
    if (sk->m > 1) c0 = kUncappedBytes; // multiplicity filter: the first stage is admitted whole (see push_device)
    if (c0 > sk->nslots / 4) c0 = sk->nslots / 4; // first chunk may admit every position
    sk->next_chunk_bytes = c0;
    sk->settled = false;
    sk->settled_total = 0;
    sk->bytes_pushed = 0;
    sk->hash_ms = 0.0;
    sk->launches = 0;
    return MHX_OK;
}

static int create_sketcher(int k, uint32_t s, uint32_t min_mult, uint64_t expected_bytes, uint64_t table_scale, mhx_sketcher **out)
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
    // table: >= 2^22 slots, >= 256 slots per sketch entry (worst-case admissions of the
    // occurrence bound at load 1; real inputs repeat their k-mers and stay far below)
    uint64_t want = (uint64_t)s * 256;
    if (want < (1ull << 22)) want = 1ull << 22;
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
    A((void **)&sk->d_tickets, kMaxLaunchesPerPush * sizeof(uint32_t));
    A((void **)&sk->d_out_keys, sk->out_cap * sizeof(uint64_t));
    A((void **)&sk->d_out_cnts, sk->out_cap * sizeof(uint32_t));
    A((void **)&sk->d_out_n, sizeof(uint32_t));
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
    if (*T < sk->last_T && sk->m > 1) sk->established = true; // only the tighten pass lowers T between host writes
    sk->last_T = *T;
    sk->occupied = occ_solid[0];
    sk->solid = occ_solid[1];
    return MHX_OK;
}

extern "C" int mhx_sketcher_push_device(mhx_sketcher *sk, const void *d_bytes, uint64_t n, int fmt)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || (!d_bytes && n)) return fail(MHX_E_ARG, "null argument");
    if (fmt != MHX_FMT_SEQ && fmt != MHX_FMT_FASTQ4) return fail(MHX_E_ARG, "unknown stream format %d", fmt);
    if (n == 0) return MHX_OK;
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
    const uint64_t ntiles64 = (a.end + kTileBytes - 1) / kTileBytes;
    if (ntiles64 > 0x7FFFFFFFull) return fail(MHX_E_ARG, "span too large for one push (%llu bytes)", (unsigned long long)n);
    const uint32_t ntiles = (uint32_t)ntiles64;
    // more input than the settling decision assumed: tighten again before admitting it wholesale
    if (sk->settled && sk->bytes_pushed + n > 2 * sk->settled_total) sk->settled = false;
    if (fmt == MHX_FMT_FASTQ4) {
        if (sk->tile_state_cap < ntiles) {
            HIPCHK(hipStreamSynchronize(g.stream));
            hipFree(sk->d_tile_state);
            sk->d_tile_state = nullptr;
            sk->tile_state_cap = 0;
            HIPCHK(hipMalloc((void **)&sk->d_tile_state, (size_t)ntiles * sizeof(uint64_t)));
            sk->tile_state_cap = ntiles;
        }
        HIPCHK(hipMemsetAsync(sk->d_tile_state, 0, (size_t)ntiles * sizeof(uint64_t), g.stream));
        HIPCHK(hipMemsetAsync(sk->d_tickets, 0, kMaxLaunchesPerPush * sizeof(uint32_t), g.stream));
    }
    a.tile_state = sk->d_tile_state;
    TableArgs ta = table_args(sk);
    if (sk->nslots >= (1ull << 23) && !getenv("MHX_EXACT_TIGHTEN")) ta.sample = 8; // big tables: sampled passes between chunks (finish() counts exactly)
    const uint64_t pushed_before = sk->bytes_pushed;
    uint32_t tile = 0;
    int launch = 0;
    while (tile < ntiles) {
        uint32_t take = ntiles - tile;
        const bool last_slot = launch == kMaxLaunchesPerPush - 1;
        if (!sk->settled && !last_slot) {
            uint64_t chunk_bytes = sk->next_chunk_bytes;
            if (sk->m > 1 && !sk->established) {
                // stages of the capped phase are defined on the bytes actually seen (pushes may be of any size):
                // the uncapped first MiB, then never more than x4 cumulative growth per launch
                const uint64_t rest_of_prefix = sk->bytes_pushed < kUncappedBytes ? kUncappedBytes - sk->bytes_pushed : 0;
                chunk_bytes = std::max<uint64_t>(rest_of_prefix, 3 * sk->bytes_pushed);
            }
            const uint64_t chunk_tiles = std::max<uint64_t>(1, chunk_bytes / kTileBytes);
            if (chunk_tiles < take) take = (uint32_t)chunk_tiles;
        }
        if (sk->m > 1 && !sk->established) {
            // Multiplicity filter: T cannot follow the data before s hashes with count >= m exist, and until
            // then every admitted k-mer costs two atomics and may be a new table entry.  The first MiB is
            // admitted whole (small genomes and saturated k-mer spaces show their solid hashes there); after
            // that T is capped at 48*s' / (bytes seen after this launch), s' = s + 8*sqrt(s) + 16, i.e. ~20*s
            // admissions per x4 stage.  The cap stays above the final s-th solid hash for any genome size while
            // the error-free k-mer coverage c so far is <= ~17x, and s solid hashes appear below it as soon as
            // c / P[Poisson(c) >= m] <= 17 (c in 0.8 .. 16 for m = 3), a window no x4 stage can jump over.
            // Inputs with fewer than s solid k-mers in total, or m > ~8, end in finish()'s exactness check and
            // the retry with a 16x budget.
            // No cap while the input looks like a small genome sequenced deeply (a fifth of the table entries
            // are solid already, yet fewer than s of them): its sketch may need every solid hash there is.
            const uint64_t after = pushed_before + std::min<uint64_t>(n, (uint64_t)(tile + take) * kTileBytes);
            const bool saturating = sk->occupied > 0 && sk->solid * 5 >= sk->occupied;
            if (after > kUncappedBytes && !saturating) {
                const long double s_eff = (long double)sk->s + 8.0L * sqrtl((long double)sk->s) + 16.0L;
                const long double t_frac = (long double)sk->last_T / (long double)sk->hash_max;
                const long double cap_frac = 48.0L * s_eff * (long double)sk->admit_scale / (long double)after;
                if (cap_frac < t_frac) {
                    sk->t_write = (uint64_t)(cap_frac * (long double)sk->hash_max);
                    HIPCHK(hipMemcpyAsync(sk->d_thresh, &sk->t_write, sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
                    HIPCHK(hipStreamSynchronize(g.stream)); // t_write is reused by the next launch
                    sk->last_T = sk->t_write;
                    sk->bounded = true;
                }
            }
        }
        a.tile0 = tile;
        a.ntiles = take;
        a.ticket = sk->d_tickets + launch;
        if (g.profiling) HIPCHK(hipEventRecord(g.ev0, g.stream));
        HIPCHK(launch_hash(sk->k, fmt, a, g.stream));
        if (g.profiling) {
            HIPCHK(hipEventRecord(g.ev1, g.stream));
            HIPCHK(hipEventSynchronize(g.ev1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, g.ev0, g.ev1));
            sk->hash_ms += ms;
        }
        ++sk->launches;
        ++launch;
        tile += take;
        sk->bytes_pushed = pushed_before + std::min<uint64_t>(n, (uint64_t)tile * kTileBytes); // real bytes, not whole tiles: callers may push tiny spans
        if (!sk->settled) {
            // tighten T from what has been seen, then decide whether the rest can go at once:
            // expected admissions of everything still to come must fit an eighth of the table
            HIPCHK(launch_tighten(ta, g.stream));
            uint64_t T;
            rc = read_threshold(sk, &T);
            if (rc) return rc;
            const uint64_t total = sk->expected_bytes > sk->bytes_pushed ? sk->expected_bytes : (uint64_t)ntiles * kTileBytes;
            const uint64_t remaining = total > sk->bytes_pushed ? total - sk->bytes_pushed : (uint64_t)(ntiles - tile) * kTileBytes;
            const long double admit = (long double)remaining * ((long double)T / (long double)sk->hash_max);
            // (with a multiplicity filter only once T comes from solid hashes: a host-imposed cap must keep
            // following the input in x4 stages, or it would drop below the final s-th solid hash)
            const bool may_settle = sk->m == 1 || sk->established;
            if (may_settle && admit <= (long double)(sk->nslots / 8)) { sk->settled = true; sk->settled_total = total > sk->bytes_pushed ? total : sk->bytes_pushed; }
            else sk->next_chunk_bytes *= (sk->m > 1 && !sk->established) ? 4 : kChunkGrowth; // x4 stages until solid hashes exist
        }
    }
    return MHX_OK;
}

extern "C" int mhx_sketcher_push_host(mhx_sketcher *sk, const void *h_bytes, uint64_t n, int fmt)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || (!h_bytes && n)) return fail(MHX_E_ARG, "null argument");
    if (n == 0) return MHX_OK;
    // the staging buffer is reused: wait for earlier pushes that may still read it
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
    (void)sk;
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
        HIPCHK(launch_extract(table_args(sk), limit, min_count, sk->d_out_keys, sk->d_out_cnts, sk->out_cap, sk->d_out_n, nullptr, nullptr, g.stream));
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

static void sort_pairs(std::vector<uint64_t> &keys, std::vector<uint32_t> &cnts)
{ // The extracted hashes are (close to) uniform below the threshold: one scatter into n buckets by value, then an
  // insertion sort over the almost-sorted result (O(n) expected; any input still ends up sorted).
    const size_t n = keys.size();
    if (n < 2) return;
    uint64_t hi = 0;
    for (size_t i = 0; i < n; ++i) hi = keys[i] > hi ? keys[i] : hi;
    std::vector<uint32_t> start(n + 1, 0);
    auto bucket = [&](uint64_t k) { return (size_t)(((unsigned __int128)k * n) / ((unsigned __int128)hi + 1)); };
    for (size_t i = 0; i < n; ++i) ++start[bucket(keys[i]) + 1];
    for (size_t b = 0; b < n; ++b) start[b + 1] += start[b];
    std::vector<uint64_t> k2(n);
    std::vector<uint32_t> c2(n);
    for (size_t i = 0; i < n; ++i) {
        const size_t d = start[bucket(keys[i])]++;
        k2[d] = keys[i];
        c2[d] = cnts[i];
    }
    for (size_t i = 1; i < n; ++i) {
        const uint64_t k = k2[i];
        const uint32_t c = c2[i];
        size_t j = i;
        while (j > 0 && k2[j - 1] > k) { k2[j] = k2[j - 1]; c2[j] = c2[j - 1]; --j; }
        k2[j] = k;
        c2[j] = c;
    }
    keys.swap(k2);
    cnts.swap(c2);
}

static int mhx_sketcher_finish_impl(mhx_sketcher *sk, uint64_t *hashes, uint32_t *counts, uint32_t *n_out)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !hashes || !n_out) return fail(MHX_E_ARG, "null argument");
    // One batch on the stream, one synchronisation: final tighten, extract with the threshold read on the
    // device, then threshold, counters, entry count and the first entries come back together.
    uint64_t T = 0;
    uint32_t n = 0;
    std::vector<uint64_t> hs(kStatReplicas * kStatCount);
    const uint32_t first = std::min<uint32_t>(sk->out_cap, 2 * sk->s + 4096);
    std::vector<uint64_t> keys(first);
    std::vector<uint32_t> cnts(first);
    HIPCHK(launch_tighten(table_args(sk), g.stream));
    HIPCHK(hipMemsetAsync(sk->d_out_n, 0, sizeof(uint32_t), g.stream));
    HIPCHK(launch_extract(table_args(sk), 0, sk->m, sk->d_out_keys, sk->d_out_cnts, sk->out_cap, sk->d_out_n, nullptr, sk->d_thresh, g.stream));
    HIPCHK(hipMemcpyAsync(&T, sk->d_thresh, sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(hs.data(), sk->d_stats, hs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(&n, sk->d_out_n, sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(keys.data(), sk->d_out_keys, (size_t)first * sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(cnts.data(), sk->d_out_cnts, (size_t)first * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    sk->last_T = T;
    uint64_t flags = 0, maxkey = 0;
    for (int r = 0; r < kStatReplicas; ++r) { flags |= hs[r * kStatCount + kStatFlags]; maxkey += hs[r * kStatCount + kStatMaxKey]; }
    rc = check_flags(flags);
    if (rc) return rc;
    if (n > first) { // more entries below T than the first copy covered (or than the device buffer holds)
        rc = extract(sk, T, sk->m, keys, cnts);
        if (rc) return rc;
    } else {
        keys.resize(n);
        cnts.resize(n);
    }
    if (T == ~0ull && maxkey >= sk->m) { // the one hash value the table cannot hold
        keys.push_back(~0ull);
        cnts.push_back((uint32_t)maxkey);
    }
    // exactness: either nothing was ever rejected, or at least s qualifying hashes lie below T
    if (keys.size() < sk->s && sk->bounded)
        return fail(MHX_E_CAPACITY, "admission threshold was too tight for this input (%zu of %u sketch entries); recreate the sketcher with a larger table",
                    keys.size(), sk->s);
    sort_pairs(keys, cnts);
    const uint32_t nn = keys.size() < sk->s ? (uint32_t)keys.size() : sk->s;
    memcpy(hashes, keys.data(), (size_t)nn * sizeof(uint64_t));
    if (counts) memcpy(counts, cnts.data(), (size_t)nn * sizeof(uint32_t));
    *n_out = nn;
    return MHX_OK;
}

static int mhx_sketcher_export_impl(mhx_sketcher *sk, uint64_t limit, uint64_t *hashes, uint32_t *counts, uint32_t cap, uint32_t *n_out)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!sk || !n_out) return fail(MHX_E_ARG, "null argument");
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
        sort_pairs(keys, cnts);
        memcpy(hashes, keys.data(), keys.size() * sizeof(uint64_t));
        memcpy(counts, cnts.data(), cnts.size() * sizeof(uint32_t));
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
    uint64_t *w = (uint64_t *)d_slab;
    HIPCHK(hipMemsetAsync(w, 0, 3 * sizeof(uint64_t), g.stream));
    HIPCHK(hipMemcpyAsync(w + 1, sk->d_thresh, sizeof(uint64_t), hipMemcpyDeviceToDevice, g.stream));
    HIPCHK(launch_extract(table_args(sk), 0, 1, w + 3, (uint32_t *)(w + 3 + cap), cap, (uint32_t *)w, w + 2, sk->d_thresh, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return MHX_OK;
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

// ---- batched distance ------------------------------------------------------------------
extern "C" double mhx_last_dist_kernel_ms(void) { return g.last_dist_ms; }

extern "C" int mhx_dist_batch(const uint64_t *q, const uint32_t *q_len, uint32_t nq, const uint64_t *r, const uint32_t *r_len,
                              uint32_t nr, uint32_t stride, int k, uint32_t s, uint32_t *common, uint32_t *denom, double *dist,
                              int device_ptrs)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (nq == 0 || nr == 0) return MHX_OK;
    if (!q || !q_len || !r || !r_len || !common || !denom) return fail(MHX_E_ARG, "null argument");
    if (k < 1 || k > 32 || s == 0 || stride == 0) return fail(MHX_E_ARG, "bad k / s / stride");
    const uint64_t pairs = (uint64_t)nq * nr;
    if (pairs > 0x7FFFFFFFull) return fail(MHX_E_ARG, "too many pairs for one call");
    DistArgs a;
    a.nq = nq; a.nr = nr; a.stride = stride; a.s = s; a.k = k; a.out_stride = nr; a.out_off = 0;
    void *dq = nullptr, *dr = nullptr, *dql = nullptr, *drl = nullptr, *dc = nullptr, *dd = nullptr, *dx = nullptr;
    auto cleanup = [&]() { hipFree(dq); hipFree(dr); hipFree(dql); hipFree(drl); hipFree(dc); hipFree(dd); hipFree(dx); };
    if (device_ptrs) {
        a.q = q; a.q_len = q_len; a.r = r; a.r_len = r_len; a.common = common; a.denom = denom; a.dist = dist;
    } else {
        for (uint32_t i = 0; i < nq; ++i) if (q_len[i] > stride) return fail(MHX_E_ARG, "q_len[%u] exceeds stride", i);
        for (uint32_t i = 0; i < nr; ++i) if (r_len[i] > stride) return fail(MHX_E_ARG, "r_len[%u] exceeds stride", i);
        hipError_t e = hipSuccess;
        auto A = [&](void **p, size_t n) { if (e == hipSuccess) e = hipMalloc(p, n); };
        A(&dq, (size_t)nq * stride * 8); A(&dr, (size_t)nr * stride * 8); A(&dql, (size_t)nq * 4); A(&drl, (size_t)nr * 4);
        A(&dc, pairs * 4); A(&dd, pairs * 4);
        if (e != hipSuccess) { cleanup(); return fail(MHX_E_HIP, "hipMalloc failed in dist_batch: %s", hipGetErrorString(e)); }
        hipError_t ce = hipMemcpyAsync(dq, q, (size_t)nq * stride * 8, hipMemcpyHostToDevice, g.stream);
        if (ce == hipSuccess) ce = hipMemcpyAsync(dr, r, (size_t)nr * stride * 8, hipMemcpyHostToDevice, g.stream);
        if (ce == hipSuccess) ce = hipMemcpyAsync(dql, q_len, (size_t)nq * 4, hipMemcpyHostToDevice, g.stream);
        if (ce == hipSuccess) ce = hipMemcpyAsync(drl, r_len, (size_t)nr * 4, hipMemcpyHostToDevice, g.stream);
        if (ce != hipSuccess) { cleanup(); return fail(MHX_E_HIP, "H2D copy failed in dist_batch: %s", hipGetErrorString(ce)); }
        a.q = (const uint64_t *)dq; a.q_len = (const uint32_t *)dql; a.r = (const uint64_t *)dr; a.r_len = (const uint32_t *)drl;
        a.common = (uint32_t *)dc; a.denom = (uint32_t *)dd; a.dist = nullptr; // distances in host libm below
    }
    // all-vs-refs fast path: the references go through in slices of 32 (one bit each in the range kernel's
    // masks), every slice filling its columns of the [nq][nr] outputs; the generic pair-per-workgroup kernel
    // serves tiny batches and is the fallback of a slice whose value ranges are too uneven for the LDS table
    const bool fast = pairs >= 64 && getenv("MHX_DIST_GENERIC") == nullptr;
    DistWork w{};
    if (fast) {
        size_t oq, orr, oc, op;
        const size_t need = dist_work_bytes(nq, nr < 32 ? nr : 32, &oq, &orr, &oc, &op);
        if (g.dist_ws_cap < need) {
            hipFree(g.dist_ws);
            g.dist_ws = nullptr;
            g.dist_ws_cap = 0;
            if (hipMalloc((void **)&g.dist_ws, need) != hipSuccess) { cleanup(); return fail(MHX_E_HIP, "hipMalloc failed for the distance workspace"); }
            g.dist_ws_cap = need;
        }
        w.offs_q = (uint32_t *)(g.dist_ws + oq); w.offs_r = (uint32_t *)(g.dist_ws + orr);
        w.cpart = (uint16_t *)(g.dist_ws + oc); w.params = (uint32_t *)(g.dist_ws + op);
    }
    hipEventRecord(g.ev0, g.stream);
    hipError_t le = hipSuccess;
    if (!fast) le = launch_dist_pairs(a, g.stream);
    for (uint32_t r0 = 0; fast && r0 < nr && le == hipSuccess; r0 += 32) {
        DistArgs slice = a;
        slice.r = a.r + (uint64_t)r0 * stride;
        slice.r_len = a.r_len + r0;
        slice.nr = nr - r0 < 32 ? nr - r0 : 32;
        slice.out_off = r0;
        le = launch_dist_ranges(slice, w, g.stream);
        if (le != hipSuccess) break;
        uint32_t flag = 0;
        hipMemcpyAsync(&flag, w.params + 1, 4, hipMemcpyDeviceToHost, g.stream);
        hipStreamSynchronize(g.stream);
        if (flag) le = launch_dist_pairs(slice, g.stream); // a value range overflowed the LDS table
    }
    hipEventRecord(g.ev1, g.stream);
    if (le != hipSuccess) { cleanup(); return fail(MHX_E_HIP, "dist kernel launch failed: %s", hipGetErrorString(le)); }
    hipError_t se = hipSuccess;
    if (!device_ptrs) {
        se = hipMemcpyAsync(common, dc, pairs * 4, hipMemcpyDeviceToHost, g.stream);
        if (se == hipSuccess) se = hipMemcpyAsync(denom, dd, pairs * 4, hipMemcpyDeviceToHost, g.stream);
    }
    if (se == hipSuccess) se = hipStreamSynchronize(g.stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, g.ev0, g.ev1);
    g.last_dist_ms = ms;
    cleanup();
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

// ---- file level -------------------------------------------------------------------------
static int put_text(const std::string &t, char *buf, size_t cap, size_t *need)
{
    if (need) *need = t.size() + 1;
    if (cap == 0) return MHX_OK;
    if (!buf || cap < t.size() + 1) return fail(MHX_E_CAPACITY, "text buffer too small (%zu needed)", t.size() + 1);
    memcpy(buf, t.c_str(), t.size() + 1);
    return MHX_OK;
}

static std::string make_comment(const std::string &name, const std::string &comment, uint64_t count)
{ // mash sketchFile(): "<name> <comment>", wrapped when several records were counted
    std::string c = name + " " + comment;
    if (count > 1) c = "[" + std::to_string(count) + " seqs] " + c + " [...]";
    return c;
}

struct Loaded {
    std::vector<uint8_t> raw;  // inflated file
    ParsedRecords rec;         // filled when the record parser path is used
    bool fastq4 = false;
};

// One reference from one or more inputs on the device.  On a 4-line violation or a too
// tight admission bound the whole reference is redone (record parser / bigger table).
static int sketch_reference(const std::vector<Loaded *> &inputs, int k, uint32_t s, uint32_t m, bool allow_device_fastq,
                            std::vector<uint64_t> &hashes, std::vector<uint32_t> &counts, uint64_t *kmers)
{
    uint64_t total = 0;
    for (auto *in : inputs) total += in->raw.size();
    bool device_fastq = allow_device_fastq;
    uint64_t boost = 1;
    for (int attempt = 0; attempt < 6; ++attempt) {
        mhx_sketcher *sk = nullptr;
        int rc = create_sketcher(k, s, m, total, boost, &sk);
        if (rc) return rc;
        for (auto *in : inputs) {
            if (device_fastq && in->fastq4) {
                rc = mhx_sketcher_push_host(sk, in->raw.data(), in->raw.size(), MHX_FMT_FASTQ4);
            } else {
                if (in->rec.records_seen == 0 && in->rec.seq.empty()) {
                    rc = parse_fastx(in->raw.data(), in->raw.size(), k, in->rec);
                    if (rc) break;
                }
                rc = mhx_sketcher_push_host(sk, in->rec.seq.data(), in->rec.seq.size(), MHX_FMT_SEQ);
            }
            if (rc) break;
        }
        uint32_t n = 0;
        if (!rc) {
            hashes.resize(s);
            counts.resize(s);
            rc = mhx_sketcher_finish(sk, hashes.data(), counts.data(), &n);
        }
        if (!rc && kmers) {
            uint64_t st[8];
            rc = mhx_sketcher_stats(sk, st);
            *kmers = st[0];
        }
        mhx_sketcher_destroy(sk);
        if (rc == MHX_E_FORMAT && device_fastq) { device_fastq = false; continue; }
        if (rc == MHX_E_CAPACITY) { boost *= 16; continue; }
        if (rc) return rc;
        hashes.resize(n);
        counts.resize(n);
        return MHX_OK;
    }
    return fail(MHX_E_CAPACITY, "could not size the device table for this input");
}

// ---- streaming FASTQ ingest ---------------------------------------------------------------
// Reads mode on real inputs is inflate-bound (zlib, ~0.1-0.3 GB/s per stream), so every input
// file gets its own inflate thread; each thread cuts its stream into <= 32 MiB chunks at record
// boundaries (a multiple of four lines since the start of the file) and hands them to the
// caller's thread, which copies them to the device and pushes them through the FASTQ kernel
// while the other files keep inflating.  Host memory stays bounded (a few chunks per file).
namespace {
constexpr size_t kIngestChunk = 32u << 20;

// One record-aligned piece of an inflated FASTQ.  The buffer keeps GzInflater::kWindow bytes of room in
// front of the data: the previous 32 KiB of the stream, which DEFLATE matches may still refer to.
struct IngestChunk {
    std::unique_ptr<uint8_t[]> buf; // kWindow + kIngestChunk + slack bytes, not zero-filled
    size_t size = 0;
    int file = 0;
    bool first_of_file = false;
    uint8_t *data() { return buf.get() + GzInflater::kWindow; }
    static size_t alloc_bytes() { return GzInflater::kWindow + kIngestChunk + GzInflater::kOvershoot + 64; }
};

class ChunkQueue {
  public:
    void put(IngestChunk &&c)
    {
        std::unique_lock<std::mutex> lk(m_);
        room_.wait(lk, [&] { return q_.size() < 4 || abort_; });
        if (abort_) return;
        q_.push_back(std::move(c));
        ready_.notify_one();
    }
    bool get(IngestChunk &out) // false when every producer is done and the queue is empty
    {
        std::unique_lock<std::mutex> lk(m_);
        ready_.wait(lk, [&] { return !q_.empty() || live_ == 0; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        room_.notify_one();
        return true;
    }
    // chunk buffers go round: fresh 32 MiB allocations cost more in page faults than the inflate that fills them
    std::unique_ptr<uint8_t[]> take_buffer()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            if (!spare_.empty()) { std::unique_ptr<uint8_t[]> b = std::move(spare_.back()); spare_.pop_back(); return b; }
        }
        return std::unique_ptr<uint8_t[]>(new uint8_t[IngestChunk::alloc_bytes()]);
    }
    void give_back(std::unique_ptr<uint8_t[]> b) { if (b) { std::lock_guard<std::mutex> lk(m_); spare_.push_back(std::move(b)); } }
    void producer_started() { std::lock_guard<std::mutex> lk(m_); ++live_; }
    void producer_done() { std::lock_guard<std::mutex> lk(m_); --live_; ready_.notify_all(); }
    void abort() { std::lock_guard<std::mutex> lk(m_); abort_ = true; room_.notify_all(); }
    bool aborted() { std::lock_guard<std::mutex> lk(m_); return abort_; }

  private:
    std::mutex m_;
    std::condition_variable ready_, room_;
    std::deque<IngestChunk> q_;
    std::vector<std::unique_ptr<uint8_t[]>> spare_;
    int live_ = 0;
    bool abort_ = false;
};

struct FileIngestState {
    std::string error;
    uint64_t lines = 0, bytes = 0;
    bool not_fastq4 = false;
    bool own_inflate_failed = false; // the engine's own DEFLATE decoder refused the stream
};

static bool is_gzip_file(const char *path)
{
    FILE *f = fopen(path, "rb");
    uint8_t magic[2] = {0, 0};
    if (f) { if (fread(magic, 1, 2, f) != 2) magic[0] = 0; fclose(f); }
    return magic[0] == 0x1f && magic[1] == 0x8b;
}

#if defined(__x86_64__)
__attribute__((target("avx2,popcnt"))) static size_t count_newlines_avx2(const uint8_t *p, size_t n)
{
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t c = 0, i = 0;
    for (; i + 128 <= n; i += 128) {
        const uint32_t m0 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i)), nl));
        const uint32_t m1 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i + 32)), nl));
        const uint32_t m2 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i + 64)), nl));
        const uint32_t m3 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i + 96)), nl));
        c += (size_t)__builtin_popcountll(((uint64_t)m1 << 32) | m0) + (size_t)__builtin_popcountll(((uint64_t)m3 << 32) | m2);
    }
    for (; i < n; ++i) c += p[i] == '\n';
    return c;
}
#endif

static size_t count_newlines(const uint8_t *p, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
    if (avx2) return count_newlines_avx2(p, n);
#endif
    size_t c = 0;
    for (size_t i = 0; i < n; ++i) c += p[i] == '\n';
    return c;
}

static bool read_whole_file(const char *path, std::vector<uint8_t> &out, size_t pad)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    struct stat sb;
    if (fstat(fileno(f), &sb) != 0) { fclose(f); return false; }
    out.assign((size_t)sb.st_size + pad, 0);
    const size_t got = fread(out.data(), 1, (size_t)sb.st_size, f);
    fclose(f);
    return got == (size_t)sb.st_size;
}

// Producer of one input file: inflates it (own DEFLATE decoder on the whole compressed file in memory;
// MHX_ZLIB_INFLATE=1 selects zlib's gzread instead; an uncompressed file is simply read) and cuts the
// stream into record-aligned chunks.
void inflate_fastq(const char *path, int file, bool force_zlib, ChunkQueue *q, FileIngestState *st)
{
    const bool gz = is_gzip_file(path);
    const bool own = gz && !force_zlib && !getenv("MHX_ZLIB_INFLATE");
    gzFile g = nullptr;
    FILE *plain = nullptr;
    std::vector<uint8_t> zbytes;
    GzInflater inf;
    if (own) {
        if (!read_whole_file(path, zbytes, 16)) { st->error = std::string("ERROR: could not open ") + path + " for reading"; q->producer_done(); return; }
        inf.set_input(zbytes.data(), zbytes.size() - 16);
    } else if (gz) {
        g = gzopen(path, "rb");
        if (g) gzbuffer(g, 1 << 20);
    } else {
        plain = fopen(path, "rb");
    }
    if (!own && !g && !plain) { st->error = std::string("ERROR: could not open ") + path + " for reading"; q->producer_done(); return; }
    auto close_all = [&]() { if (g) gzclose(g); if (plain) fclose(plain); };
    std::vector<uint8_t> tail; // [history in front of the carry][carry]: the end of the previous chunk's stream
    size_t carry_len = 0;
    uint64_t produced = 0;     // inflated bytes so far (bounds how far back a match may reach)
    bool first = true;
    uint64_t lines_before = 0; // newlines in everything already emitted
    const bool dbg = getenv("MHX_INGEST_DEBUG") != nullptr;
    double t_alloc = 0, t_inflate = 0, t_cut = 0, t_put = 0;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    struct Report { bool on; const char *path; double *a, *i, *c, *p; ~Report() { if (on) fprintf(stderr, "ingest %s: alloc %.3f inflate %.3f cut %.3f put-wait %.3f s\n", path, *a, *i, *c, *p); } } report{dbg, path, &t_alloc, &t_inflate, &t_cut, &t_put};
    for (;;) {
        auto t0 = now();
        IngestChunk c;
        c.file = file;
        c.first_of_file = first;
        c.buf = q->take_buffer();
        uint8_t *d = c.data();
        if (!tail.empty()) memcpy(d + carry_len - tail.size(), tail.data(), tail.size());
        t_alloc += secs(t0, now());
        t0 = now();
        size_t n = carry_len;
        bool eof = false;
        while (n < kIngestChunk) {
            long got;
            if (own) {
                const uint64_t hist = produced < GzInflater::kWindow ? produced : GzInflater::kWindow;
                const size_t r = inf.inflate(d + n, kIngestChunk - n, d + n - hist);
                if (r == (size_t)-1) got = -1;
                else { got = (long)r; produced += r; if (inf.done()) { n += r; eof = true; break; } }
            } else if (gz) {
                got = gzread(g, d + n, (unsigned)std::min<size_t>(kIngestChunk - n, 1u << 30));
            } else {
                got = (long)fread(d + n, 1, kIngestChunk - n, plain);
                if (got == 0 && ferror(plain)) got = -1;
            }
            if (got < 0) {
                st->error = std::string("ERROR: reading ") + path + " failed";
                st->own_inflate_failed = own; // the caller repeats the run with zlib before giving up
                close_all();
                q->producer_done();
                return;
            }
            if (got == 0) { eof = true; break; }
            n += (size_t)got;
        }
        t_inflate += secs(t0, now());
        t0 = now();
        if (first && n && d[0] != '@') { st->not_fastq4 = true; close_all(); q->producer_done(); return; }
        // cut after the last newline that completes a record (line count multiple of 4): a vectorised
        // newline count, then a short walk back over the unfinished last record; the records themselves
        // are parsed and counted on the device
        size_t cut = 0;
        uint64_t lines = lines_before + count_newlines(d, n), lines_at_cut = lines_before;
        {
            uint64_t back = lines & 3;
            size_t end = n;
            const uint8_t *p = (const uint8_t *)memrchr(d, '\n', end);
            while (p && back) { end = (size_t)(p - d); p = (const uint8_t *)memrchr(d, '\n', end); --back; }
            if (p) { cut = (size_t)(p - d) + 1; lines_at_cut = lines - (lines & 3); }
        }
        if (eof) {
            // the tail must be whole records; a last record may lack its final newline
            if (cut < n) { const uint64_t tail_lines = lines - lines_at_cut + 1; if (tail_lines != 4) st->not_fastq4 = true; }
            cut = n;
            lines_at_cut = lines + (n && d[n - 1] != '\n' ? 1 : 0);
        } else if (cut == 0) {
            st->not_fastq4 = true; // a single record larger than a chunk: leave it to the record parser
        }
        if (st->not_fastq4) { close_all(); q->producer_done(); return; }
        carry_len = n - cut;
        if (!eof) {
            // the next chunk starts with the carry; in front of it goes what is left of the 32 KiB window
            const size_t want = carry_len >= GzInflater::kWindow ? carry_len : GzInflater::kWindow;
            const size_t have = (size_t)std::min<uint64_t>(own ? produced : 0, want); // zlib keeps its own window
            const size_t keep = have > carry_len ? have : carry_len;
            tail.assign(d + n - keep, d + n);
        }
        c.size = cut;
        st->bytes += cut;
        st->lines = lines_at_cut;
        lines_before = lines_at_cut;
        first = false;
        t_cut += secs(t0, now());
        t0 = now();
        if (cut) q->put(std::move(c));
        t_put += secs(t0, now());
        if (eof || q->aborted()) break;
    }
    close_all();
    q->producer_done();
}

uint64_t guess_inflated_bytes(const char *path)
{
    struct stat sb;
    if (stat(path, &sb) != 0) return 0;
    return (uint64_t)sb.st_size * (is_gzip_file(path) ? 8 : 1);
}
} // namespace

// ---- bulk ingest of an uncompressed FASTQ file -------------------------------------------
// The whole file goes into ONE device buffer and ONE push: reader threads pread() disjoint
// ranges of a 64 MiB block into a pinned staging slot, the block is copied to its place in the
// device buffer on a copy stream while the next block is being read, and the device parser
// finds the records itself (line phase by look-back), so the host never scans the bytes.
namespace {
constexpr size_t kBulkBlock = 64u << 20;

struct BulkFile {
    uint8_t *d_buf = nullptr;
    uint64_t size = 0;
    std::vector<uint8_t> head; // first bytes of the file (record name / comment)
};

static int ensure_pinned_ring()
{
    if (!g.copy_stream) HIPCHK(hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < Engine::kPinnedSlots; ++i) {
        if (!g.pinned[i]) HIPCHK(hipHostMalloc((void **)&g.pinned[i], kBulkBlock, hipHostMallocDefault));
        if (!g.pinned_free[i]) HIPCHK(hipEventCreateWithFlags(&g.pinned_free[i], hipEventDisableTiming));
    }
    return MHX_OK;
}

// reads [off, off + len) of fd into dst with `nthreads` parallel preads; false on a short read
static bool parallel_pread(int fd, uint8_t *dst, uint64_t off, size_t len, int nthreads)
{
    std::vector<std::thread> th;
    std::vector<int> ok((size_t)nthreads, 1);
    const size_t per = ((len + (size_t)nthreads - 1) / (size_t)nthreads + 4095) & ~(size_t)4095;
    for (int t = 0; t < nthreads; ++t) {
        const size_t b = (size_t)t * per;
        if (b >= len) break;
        const size_t e = std::min(len, b + per);
        th.emplace_back([=, &ok]() {
            size_t done = b;
            while (done < e) {
                const ssize_t got = pread(fd, dst + done, e - done, (off_t)(off + done));
                if (got <= 0) { ok[(size_t)t] = 0; return; }
                done += (size_t)got;
            }
        });
    }
    for (auto &t : th) t.join();
    for (int v : ok) if (!v) return false;
    return true;
}

// MHX_OK and f->d_buf set, or MHX_OK with d_buf == nullptr when the file should take another path
static int bulk_load_plain(const char *path, BulkFile *f)
{
    struct stat sb;
    if (stat(path, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size <= 0) return MHX_OK;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return MHX_OK;
    if ((uint64_t)sb.st_size + (4ull << 30) > free_b / 2) return MHX_OK; // leave room for tables and other files
    int rc = ensure_pinned_ring();
    if (rc) return rc;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(MHX_E_IO, "ERROR: could not open %s for reading", path);
    f->size = (uint64_t)sb.st_size;
    if (hipMalloc((void **)&f->d_buf, f->size + 64) != hipSuccess) { close(fd); f->d_buf = nullptr; return MHX_OK; }
    int nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads > 16) nthreads = 16;
    if (nthreads < 1) nthreads = 1;
    uint64_t off = 0;
    int slot = 0;
    rc = MHX_OK;
    while (off < f->size && !rc) {
        const size_t len = (size_t)std::min<uint64_t>(kBulkBlock, f->size - off);
        if (hipEventSynchronize(g.pinned_free[slot]) != hipSuccess) { rc = fail(MHX_E_HIP, "pinned slot wait failed"); break; }
        if (!parallel_pread(fd, g.pinned[slot], off, len, len >= (8u << 20) ? nthreads : 1)) { rc = fail(MHX_E_IO, "ERROR: reading %s failed", path); break; }
        if (off == 0) f->head.assign(g.pinned[slot], g.pinned[slot] + std::min<size_t>(len, 1u << 20));
        if (hipMemcpyAsync(f->d_buf + off, g.pinned[slot], len, hipMemcpyHostToDevice, g.copy_stream) != hipSuccess ||
            hipEventRecord(g.pinned_free[slot], g.copy_stream) != hipSuccess) { rc = fail(MHX_E_HIP, "H2D copy failed"); break; }
        off += len;
        slot = (slot + 1) % Engine::kPinnedSlots;
    }
    close(fd);
    if (!rc && hipMemsetAsync(f->d_buf + f->size, 0, 64, g.copy_stream) != hipSuccess) rc = fail(MHX_E_HIP, "memset failed");
    if (!rc && hipStreamSynchronize(g.copy_stream) != hipSuccess) rc = fail(MHX_E_HIP, "copy stream sync failed");
    if (rc) { hipStreamSynchronize(g.copy_stream); hipFree(f->d_buf); f->d_buf = nullptr; }
    return rc;
}

// name / comment of the first record mash would count (sequence of at least k bytes) in `buf`; false
// when none of the records there is long enough (name / comment then hold the first header as a last resort)
static bool first_counted_header(const uint8_t *buf, size_t n, int k, std::string &name, std::string &comment)
{
    size_t p = 0;
    while (p < n) {
        const uint8_t *h_end = (const uint8_t *)memchr(buf + p, '\n', n - p);
        if (!h_end) break;
        const size_t s0 = (size_t)(h_end - buf) + 1;
        const uint8_t *s_end = s0 < n ? (const uint8_t *)memchr(buf + s0, '\n', n - s0) : nullptr;
        const size_t s1 = s_end ? (size_t)(s_end - buf) : n;
        const size_t seq_len = s1 - s0 - ((s1 > s0 && buf[s1 - 1] == '\r') ? 1 : 0); // CRLF files: the CR is not a base
        if (seq_len >= (size_t)k) { first_header(buf + p, s1 - p, name, comment); return true; }
        // skip the '+' and quality lines
        size_t q = s1 + 1;
        for (int i = 0; i < 2 && q < n; ++i) {
            const uint8_t *e = (const uint8_t *)memchr(buf + q, '\n', n - q);
            q = e ? (size_t)(e - buf) + 1 : n;
        }
        p = q;
    }
    first_header(buf, n, name, comment);
    return false;
}

// Which record names the reference: the first counted record of the lowest-numbered file that has one
// (within its first chunk); the very first header if no file has any.
struct HeaderPick {
    int file = -1;        // file that provided a counted record
    bool fallback = false;
    std::string name, comment, fb_name, fb_comment;
    void offer(int f, const uint8_t *buf, size_t n, int k)
    {
        if (file >= 0 && f > file) return;
        std::string nm, cm;
        if (first_counted_header(buf, n, k, nm, cm)) { file = f; name = nm; comment = cm; }
        else if (!fallback || f == 0) { fallback = true; fb_name = nm; fb_comment = cm; }
    }
    void result(std::string &nm, std::string &cm) const
    {
        if (file >= 0) { nm = name; cm = comment; }
        else { nm = fb_name; cm = fb_comment; }
    }
};
} // namespace

// returns MHX_OK with *handled = true when the streaming path produced the sketch;
// *handled = false means "not strict FASTQ / could not size": use the whole-file path.
static int stream_fastq_reference(const char *const *paths, int n_paths, int k, uint32_t s, uint32_t m, std::vector<uint64_t> &hashes,
                                  std::vector<uint32_t> &counts, uint64_t *kmers, uint64_t *records, std::string *fname,
                                  std::string *fcomment, bool *handled, bool force_zlib = false)
{
    *handled = false;
    uint64_t expected = 0;
    for (int i = 0; i < n_paths; ++i) expected += guess_inflated_bytes(paths[i]);
    mhx_sketcher *sk = nullptr;
    int rc = mhx_sketcher_create(k, s, m, expected, &sk);
    if (rc) return rc;
    bool fallback = false;
    HeaderPick header;
    // 1. uncompressed files: whole file -> one device buffer -> one push (see bulk_load_plain).  The buffers
    // stay on the device until the sketch is final, so that a too-small admission budget can be repaired by
    // pushing them again into a larger sketcher instead of reading the files a second time.
    std::vector<int> queued; // files that go through an inflate thread instead
    std::vector<BulkFile> resident;
    auto free_resident = [&]() { for (auto &b : resident) hipFree(b.d_buf); resident.clear(); };
    for (int i = 0; i < n_paths && !rc && !fallback; ++i) {
        if (is_gzip_file(paths[i]) || getenv("MHX_NO_BULK")) { queued.push_back(i); continue; }
        BulkFile bf;
        rc = bulk_load_plain(paths[i], &bf);
        if (rc) break;
        if (!bf.d_buf) { queued.push_back(i); continue; }
        if (bf.head.empty() || bf.head[0] != '@') fallback = true;
        if (!fallback) header.offer(i, bf.head.data(), bf.head.size(), k);
        if (!fallback) rc = mhx_sketcher_push_device(sk, bf.d_buf, bf.size, MHX_FMT_FASTQ4);
        if (hipStreamSynchronize(g.stream) != hipSuccess && !rc) rc = fail(MHX_E_HIP, "stream sync failed");
        bf.head.clear();
        resident.push_back(std::move(bf));
    }
    // 2. compressed files: one inflate thread per file, 32 MiB record-aligned chunks
    uint8_t *d_slot = nullptr;
    if (!rc && !fallback && !queued.empty() && hipMalloc((void **)&d_slot, kIngestChunk + GzInflater::kOvershoot + 64) != hipSuccess)
        rc = fail(MHX_E_HIP, "hipMalloc failed for the ingest slot");
    std::vector<FileIngestState> st(n_paths);
    if (!rc && !fallback && !queued.empty()) {
        ChunkQueue q;
        std::vector<std::thread> threads;
        for (size_t j = 0; j < queued.size(); ++j) q.producer_started();
        for (int i : queued) threads.emplace_back(inflate_fastq, paths[i], i, force_zlib, &q, &st[i]);
        IngestChunk c;
        while (q.get(c)) {
            if (rc) continue; // drain
            if (c.first_of_file) header.offer(c.file, c.data(), std::min<size_t>(c.size, 1u << 20), k);
            if (hipMemcpyAsync(d_slot, c.data(), c.size, hipMemcpyHostToDevice, g.stream) != hipSuccess) { rc = fail(MHX_E_HIP, "H2D copy failed"); q.abort(); continue; }
            rc = mhx_sketcher_push_device(sk, d_slot, c.size, MHX_FMT_FASTQ4);
            if (!rc && hipStreamSynchronize(g.stream) != hipSuccess) rc = fail(MHX_E_HIP, "stream sync failed");
            if (rc) q.abort();
            else q.give_back(std::move(c.buf)); // the copy out of it has completed
        }
        for (auto &t : threads) t.join();
    }
    bool own_failed = false;
    for (auto &f : st) own_failed = own_failed || f.own_inflate_failed;
    if (own_failed && !force_zlib) { // the engine's own decoder refused a stream: let zlib have the last word
        hipFree(d_slot);
        free_resident();
        mhx_sketcher_destroy(sk);
        clear_error();
        return stream_fastq_reference(paths, n_paths, k, s, m, hashes, counts, kmers, records, fname, fcomment, handled, true);
    }
    for (auto &f : st) {
        if (!f.error.empty() && !rc) rc = fail(MHX_E_IO, "%s", f.error.c_str());
        if (f.not_fastq4) fallback = true;
    }
    uint32_t n = 0;
    if (!rc && !fallback) {
        hashes.resize(s);
        counts.resize(s);
        rc = mhx_sketcher_finish(sk, hashes.data(), counts.data(), &n);
        // admission budget too small (few solid k-mers: the sketch needs hashes the cap rejected): when every
        // input is still resident on the device, push it again into a sketcher with 16x, 256x ... the budget
        uint32_t scale = 1;
        while (rc == MHX_E_CAPACITY && queued.empty() && !resident.empty() && scale < (1u << 20)) {
            scale *= 16;
            clear_error();
            mhx_sketcher_destroy(sk);
            sk = nullptr;
            rc = create_sketcher(k, s, m, expected, scale, &sk);
            for (size_t i = 0; i < resident.size() && !rc; ++i) rc = mhx_sketcher_push_device(sk, resident[i].d_buf, resident[i].size, MHX_FMT_FASTQ4);
            if (!rc) rc = mhx_sketcher_finish(sk, hashes.data(), counts.data(), &n);
        }
        if (rc == MHX_E_FORMAT || rc == MHX_E_CAPACITY) { fallback = true; rc = MHX_OK; clear_error(); }
    }
    if (!rc && !fallback) {
        uint64_t stt[8];
        rc = mhx_sketcher_stats(sk, stt);
        *kmers = stt[0];
        header.result(*fname, *fcomment);
        if (!rc) rc = mhx_sketcher_record_count(sk, records); // sequences of >= k bytes, counted by the device parser
        hashes.resize(n);
        counts.resize(n);
        *handled = !rc;
    }
    hipFree(d_slot);
    free_resident();
    if (sk) mhx_sketcher_destroy(sk);
    return rc;
}

static int mhx_sketch_files_impl(const char *const *paths, int n_paths, int k, uint32_t s, int reads, uint32_t min_mult,
                                const char *out_msh, char *stderr_buf, size_t stderr_cap, size_t *stderr_need,
                                double *est_genome_size)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!paths || n_paths <= 0 || !out_msh) return fail(MHX_E_ARG, "sketch: paths and output required");
    if (!hash_k_supported(k)) return fail(MHX_E_ARG, "k-mer size %d not supported (1..32)", k);
    SketchSet set;
    set.kmer_size = (uint32_t)k;
    set.sketch_size = s;
    std::string err;
    std::vector<Loaded> loaded(n_paths);
    auto no_records = [&](const char *p) {
        err += std::string("ERROR: Did not find fasta records in \"") + p + "\".\n";
        put_text(err, stderr_buf, stderr_cap, stderr_need);
        return fail(MHX_E_NO_RECORDS, "ERROR: Did not find fasta records in \"%s\".", p);
    };
    if (reads) {
        RefSketch ref;
        uint64_t kmers = 0, count = 0;
        std::string fname, fcomment;
        bool streamed = false;
        if (!getenv("MHX_NO_STREAMING")) {
            rc = stream_fastq_reference(paths, n_paths, k, s, min_mult ? min_mult : 1, ref.hashes, ref.counts, &kmers, &count, &fname, &fcomment, &streamed);
            if (rc) return rc;
        }
        std::vector<Loaded *> in;
        if (!streamed) {
            for (int i = 0; i < n_paths; ++i) {
                rc = read_all_maybe_gz(paths[i], loaded[i].raw);
                if (rc) return rc;
                loaded[i].fastq4 = looks_like_fastq4(loaded[i].raw.data(), loaded[i].raw.size());
                in.push_back(&loaded[i]);
            }
            rc = sketch_reference(in, k, s, min_mult ? min_mult : 1, true, ref.hashes, ref.counts, &kmers);
            if (rc) return rc;
        }
        // name / comment / count: first counted record; count = records seen by the parser,
        // or lines / 4 when the stream went to the device parser untouched
        bool any = streamed;
        for (auto &l : loaded) {
            if (streamed) break;
            if (l.rec.records_seen || !l.rec.seq.empty()) {
                if (!any && l.rec.records) { fname = l.rec.first_name; fcomment = l.rec.first_comment; any = true; }
                count += l.rec.records;
            } else if (!l.raw.empty()) {
                uint64_t lines = 0;
                for (size_t off = 0; off < l.raw.size();) {
                    const void *p = memchr(l.raw.data() + off, '\n', l.raw.size() - off);
                    if (!p) { ++lines; break; }
                    ++lines;
                    off = (const uint8_t *)p - l.raw.data() + 1;
                }
                if (!any) { first_header(l.raw.data(), l.raw.size(), fname, fcomment); any = true; }
                count += lines / 4;
            }
        }
        if (kmers == 0 && count == 0) return no_records(paths[0]);
        double set_size = 0.0, mult = 0.0;
        if (!ref.hashes.empty()) {
            set_size = pow(2.0, k > 16 ? 64.0 : 32.0) * (double)ref.hashes.size() / (double)ref.hashes.back();
            uint64_t sum = 0;
            for (uint32_t c : ref.counts) sum += c;
            mult = (double)sum / (double)ref.hashes.size();
        }
        ref.name = paths[0];
        ref.comment = make_comment(fname, fcomment, count);
        ref.length = (uint64_t)set_size;
        ref.counts.clear(); // mash stores counts only with -M
        set.refs.push_back(std::move(ref));
        err += "Estimated genome size: " + fmt_g(set_size) + "\n";
        err += "Estimated coverage:    " + fmt_g(mult) + "\n";
        if (est_genome_size) *est_genome_size = set_size;
    } else {
        for (int i = 0; i < n_paths; ++i) {
            err += std::string("Sketching ") + paths[i] + "...\n";
            rc = read_all_maybe_gz(paths[i], loaded[i].raw);
            if (rc) return rc;
            rc = parse_fastx(loaded[i].raw.data(), loaded[i].raw.size(), k, loaded[i].rec);
            if (rc) return rc;
            if (loaded[i].rec.records == 0) return no_records(paths[i]);
            RefSketch ref;
            std::vector<Loaded *> in{&loaded[i]};
            rc = sketch_reference(in, k, s, 1, false, ref.hashes, ref.counts, nullptr);
            if (rc) return rc;
            ref.counts.clear();
            ref.name = paths[i];
            ref.comment = make_comment(loaded[i].rec.first_name, loaded[i].rec.first_comment, loaded[i].rec.records);
            ref.length = loaded[i].rec.total_length;
            set.refs.push_back(std::move(ref));
            loaded[i] = Loaded();
        }
        if (est_genome_size) *est_genome_size = 0.0;
    }
    err += std::string("Writing to ") + out_msh + "...\n";
    rc = msh_write_file(out_msh, set);
    if (rc) return rc;
    return put_text(err, stderr_buf, stderr_cap, stderr_need);
}

static int mhx_dist_files_impl(const char *ref_msh, const char *qry_msh, char *stdout_buf, size_t cap, size_t *need)
{
    clear_error();
    int rc = require_engine();
    if (rc) return rc;
    if (!ref_msh || !qry_msh) return fail(MHX_E_ARG, "dist: two sketch paths required");
    SketchSet R, Q;
    rc = msh_read_file(ref_msh, R);
    if (rc) return rc;
    rc = msh_read_file(qry_msh, Q);
    if (rc) return rc;
    if (R.kmer_size != Q.kmer_size)
        return fail(MHX_E_MISMATCH, "ERROR: The query and reference sketches have different k-mer sizes (%u and %u)", Q.kmer_size, R.kmer_size);
    if (R.hash_seed != Q.hash_seed) return fail(MHX_E_MISMATCH, "ERROR: The query and reference sketches have different hash seeds");
    const int k = (int)R.kmer_size;
    const uint32_t s = R.sketch_size < Q.sketch_size ? R.sketch_size : Q.sketch_size;
    const uint32_t nr = (uint32_t)R.refs.size(), nq = (uint32_t)Q.refs.size();
    std::string text;
    if (nr && nq) {
        uint32_t stride = 1;
        for (auto &r : R.refs) stride = std::max<uint32_t>(stride, (uint32_t)r.hashes.size());
        for (auto &q : Q.refs) stride = std::max<uint32_t>(stride, (uint32_t)q.hashes.size());
        std::vector<uint64_t> rq((size_t)nr * stride, 0), qq((size_t)nq * stride, 0);
        std::vector<uint32_t> rl(nr), ql(nq);
        for (uint32_t i = 0; i < nr; ++i) { rl[i] = (uint32_t)R.refs[i].hashes.size(); if (rl[i]) memcpy(&rq[(size_t)i * stride], R.refs[i].hashes.data(), (size_t)rl[i] * 8); }
        for (uint32_t i = 0; i < nq; ++i) { ql[i] = (uint32_t)Q.refs[i].hashes.size(); if (ql[i]) memcpy(&qq[(size_t)i * stride], Q.refs[i].hashes.data(), (size_t)ql[i] * 8); }
        std::vector<uint32_t> common((size_t)nq * nr), denom((size_t)nq * nr);
        std::vector<double> dist((size_t)nq * nr);
        rc = mhx_dist_batch(qq.data(), ql.data(), nq, rq.data(), rl.data(), nr, stride, k, s, common.data(), denom.data(), dist.data(), 0);
        if (rc) return rc;
        for (uint32_t qi = 0; qi < nq; ++qi)
            for (uint32_t ri = 0; ri < nr; ++ri) {
                const size_t p = (size_t)qi * nr + ri;
                const double pv = mhx_p_value(common[p], R.refs[ri].length, Q.refs[qi].length, k, denom[p]);
                text += R.refs[ri].name + "\t" + Q.refs[qi].name + "\t" + fmt_g(dist[p]) + "\t" + fmt_g(pv) + "\t" +
                        std::to_string(common[p]) + "/" + std::to_string(denom[p]) + "\n";
            }
    }
    return put_text(text, stdout_buf, cap, need);
}

extern "C" int mhx_msh_write(const char *path, int k, uint32_t s, uint32_t n_refs, const char *const *names,
                             const char *const *comments, const uint64_t *lengths, const uint64_t *const *hashes,
                             const uint32_t *n_hashes)
{
    clear_error();
    if (!path || (n_refs && (!names || !comments || !lengths || !hashes || !n_hashes))) return fail(MHX_E_ARG, "null argument");
    SketchSet set;
    set.kmer_size = (uint32_t)k;
    set.sketch_size = s;
    set.refs.resize(n_refs);
    for (uint32_t i = 0; i < n_refs; ++i) {
        set.refs[i].name = names[i] ? names[i] : "";
        set.refs[i].comment = comments[i] ? comments[i] : "";
        set.refs[i].length = lengths[i];
        if (n_hashes[i]) set.refs[i].hashes.assign(hashes[i], hashes[i] + n_hashes[i]);
    }
    return msh_write_file(path, set);
}

extern "C" int mhx_sketch_files(const char *const *paths, int n_paths, int k, uint32_t s, int reads, uint32_t min_mult,
                                const char *out_msh, char *stderr_buf, size_t stderr_cap, size_t *stderr_need,
                                double *est_genome_size)
{
    try {
        return mhx_sketch_files_impl(paths, n_paths, k, s, reads, min_mult, out_msh, stderr_buf, stderr_cap, stderr_need, est_genome_size);
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_sketch_files: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_sketch_files: %s", e.what());
    }
}

extern "C" int mhx_dist_files(const char *ref_msh, const char *qry_msh, char *stdout_buf, size_t cap, size_t *need)
{
    try {
        return mhx_dist_files_impl(ref_msh, qry_msh, stdout_buf, cap, need);
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_dist_files: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_dist_files: %s", e.what());
    }
}

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
