// mhx_pinflate.cpp -- one gzip member decoded by several threads.
//
// A `.fq.gz` is one long DEFLATE stream (auriclass/classes.py:588 passes it to `mash sketch`, whose kseq reads it
// through zlib, one thread).  Once the sketch runs on the GPU that single decoding thread is what a sample waits
// for, so the FASTQ ingest cuts the compressed stream into segments and decodes them side by side:
//
//   1. block search   a segment may start at any bit.  From its target offset on, every bit position is tried as a
//                     block header (dynamic or fixed Huffman); a candidate counts once its whole block decodes and a
//                     second valid header follows.
//   2. symbolic pass  the 32 KiB in front of a segment are unknown while its predecessor is still running, so the
//                     segment decodes into 16-bit symbols: a byte, or a marker 0x8000 | w for "byte w of the unknown
//                     window" (the window is laid in front of the output as 32768 markers, which makes every match a
//                     plain copy).  Whenever the last 32 KiB of output hold no marker any more the rest of the segment
//                     goes through the ordinary byte decoder.  A segment ends exactly on the block boundary its
//                     successor starts at -- anything else means a false start and fails the run.
//   3. resolution     when the predecessor's last 32 KiB are known the markers are replaced (its own last 32 KiB first,
//                     so the chain moves on; then the rest), the segment's CRC-32 is taken, and the consumer gets the
//                     bytes in order; segment CRCs are joined with zlib's crc32_combine and checked against the trailer.
//
// The result is bit-identical to the sequential decoder's or the run fails (the caller then repeats with zlib, as it
// does for any stream the own decoder refuses).  tests/test_lib_cpu.py runs the differential tests against zlib.
#include <stdio.h>
#include <stdlib.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "mhx_inflate_impl.h"
#include "mhx_internal.h"

namespace mhx {

using namespace deflate;

namespace {

constexpr uint32_t kWin = 32768;
constexpr uint16_t kMarker = 0x8000;

// malloc'ed and never zero-filled; recycled through a pool (fresh pages fault in more slowly than they are decoded into)
struct RawBuf {
    uint8_t *p = nullptr;
    size_t cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    RawBuf(RawBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    RawBuf &operator=(RawBuf &&o) noexcept { if (this != &o) { free(p); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    ~RawBuf() { free(p); }
    void ensure(size_t n) // keeps the content
    {
        if (n <= cap) return;
        size_t want = cap ? cap : (1u << 20);
        while (want < n) want += want / 2 + 4096;
        void *q = realloc(p, want);
        if (!q) throw std::bad_alloc();
        p = (uint8_t *)q;
        cap = want;
    }
};

struct Segment {
    uint64_t start_bit = 0, stop_bit = 0; // stop_bit == ~0: runs to the final block
    RawBuf sym_buf;                       // uint16_t: [kWin markers][symbols...]
    size_t n_sym = 0;                     // symbols decoded symbolically
    RawBuf out_buf;                       // [n_sym bytes to be resolved][bytes of the plain pass]
    uint16_t *sym() { return reinterpret_cast<uint16_t *>(sym_buf.p); }
    size_t sym_cap() const { return sym_buf.cap / 2; }
    uint8_t *out() { return out_buf.p; }
    size_t n_out = 0;                     // total bytes of the segment
    uint64_t end_bit = 0;                 // bit position after the segment's last block
    bool final_block_seen = false;
    uint32_t crc = 0;
    // hand-off
    bool decoded = false, window_ready = false, done = false;
    std::atomic<bool> failed{false};      // set by the worker that owns the segment (error first), read by its neighbours
    uint8_t window[kWin];                 // last 32 KiB of this segment's output (resolved)
    uint32_t window_len = 0;              // < kWin only for the first segment of a short stream
    std::string error;
};

bool any_marker(const uint16_t *p, size_t n)
{
    uint64_t acc = 0;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        uint64_t v;
        memcpy(&v, p + i, 8);
        acc |= v;
    }
    for (; i < n; ++i) acc |= p[i];
    return (acc & 0x8000800080008000ull) != 0;
}

// The consumer of a decoder copies every decoded byte once more, from the segment (or block group) it was decoded into
// to where the caller wants it -- on ONE thread that was what bounded a file's rate (~8 GB/s, with 31 decoding threads
// waiting for it).  Two helper threads that exist for the lifetime of the reader take a third of every large copy each.
class CopyTeam {
  public:
    CopyTeam() = default;
    ~CopyTeam()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void copy(uint8_t *dst, const uint8_t *src, size_t n)
    {
        if (n < (1u << 20)) { memcpy(dst, src, n); return; }
        if (!started_) { // the helpers appear with the first large copy (a reader that is never read from costs no threads)
            started_ = true;
            try {
                for (int i = 0; i < kHelpers; ++i) th_.emplace_back([this, i] { run(i); });
            } catch (const std::system_error &) { // no more threads to be had: the caller's thread copies alone, or with one helper
            }
        }
        const size_t helpers = th_.size();
        if (helpers == 0) { memcpy(dst, src, n); return; }
        const size_t part = (n / (helpers + 1) + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(m_);
            dst_ = dst; src_ = src; n_ = n; part_ = part;
            pending_ = (int)helpers;
            ++gen_;
        }
        cv_.notify_all();
        memcpy(dst, src, std::min(part, n)); // this thread's share: the first part
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
    }

  private:
    static constexpr int kHelpers = 2;
    void run(int idx)
    {
        uint64_t seen = 0;
        for (;;) {
            uint8_t *dst; const uint8_t *src; size_t n, part;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                dst = dst_; src = src_; n = n_; part = part_;
            }
            const size_t b = std::min(n, part * (size_t)(idx + 1)), e = std::min(n, part * (size_t)(idx + 2));
            // (the last helper takes what the rounding left over)
            const size_t end = idx == (int)th_.size() - 1 ? n : e;
            if (end > b) memcpy(dst + b, src + b, end - b);
            {
                std::lock_guard<std::mutex> lk(m_);
                --pending_;
            }
            done_.notify_one();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    uint8_t *dst_ = nullptr;
    const uint8_t *src_ = nullptr;
    size_t n_ = 0, part_ = 0;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false, started_ = false;
};

} // namespace

struct ParallelGunzip::Impl {
    const uint8_t *z = nullptr;
    size_t n = 0;
    size_t deflate_off = 0; // first byte of the member's deflate stream
    int nthreads = 1;
    std::vector<std::unique_ptr<Segment>> segs;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<size_t> next_task{0};
    size_t consumed = 0;     // segments the consumer is done with (guards the look-ahead)
    size_t lookahead = 8;
    std::atomic<bool> abort{false};
    // memory bound: bytes of decoded segments the consumer has not taken yet (16-bit symbols + bytes = 3 per output
    // byte); a worker does not START a further segment while this exceeds the budget -- except the one the consumer
    // waits for, so the pipeline cannot stall
    size_t outstanding = 0;
    size_t budget = (size_t)768 << 20;
    // consumer state
    size_t cur = 0, cur_off = 0;
    uint32_t crc_all = 0;
    uint64_t total_out = 0;
    bool finished = false, failed = false;
    size_t trailer_off = 0; // byte offset of the member trailer, once the last segment is in
    std::string error;
    std::vector<RawBuf> pool_out, pool_sym;
    CopyTeam copier; // read(): the consumer's copies
    RawBuf take(std::vector<RawBuf> &pool)
    {
        std::lock_guard<std::mutex> lk(m);
        if (pool.empty()) return RawBuf();
        RawBuf b = std::move(pool.back());
        pool.pop_back();
        return b;
    }
    void give(std::vector<RawBuf> &pool, RawBuf &&b)
    {
        if (!b.p) return;
        std::lock_guard<std::mutex> lk(m);
        pool.push_back(std::move(b));
    }

    ~Impl()
    {
        { std::lock_guard<std::mutex> lk(m); abort = true; }
        cv.notify_all();
        for (auto &t : workers) if (t.joinable()) t.join();
    }
    bool find_block_start(uint64_t from_bit, uint64_t to_bit, uint64_t *found) const;
    void decode_segment(size_t j);
    void worker();
};

// gzip member header at z[0..): returns the offset of the deflate stream or 0
static size_t member_header_len(const uint8_t *z, size_t n)
{
    if (n < 18 || z[0] != 0x1f || z[1] != 0x8b || z[2] != 8) return 0;
    const uint8_t flg = z[3];
    size_t p = 10;
    if (flg & 4) {
        if (n - p < 2) return 0;
        const size_t xlen = z[p] | (z[p + 1] << 8);
        p += 2;
        if (n - p < xlen) return 0;
        p += xlen;
    }
    for (int bit = 8; bit <= 16; bit <<= 1) {
        if (!(flg & bit)) continue;
        const void *e = memchr(z + p, 0, n - p);
        if (!e) return 0;
        p = (size_t)((const uint8_t *)e - z) + 1;
    }
    if (flg & 2) { if (n - p < 2) return 0; p += 2; }
    return p < n ? p : 0;
}

// First bit position in [from_bit, to_bit) at which a non-final dynamic-Huffman block starts: its header parses (a complete
// code-length code, valid length runs, an end-of-block symbol), the whole block decodes (symbolically) and a second
// dynamic header follows.
bool ParallelGunzip::Impl::find_block_start(uint64_t from_bit, uint64_t to_bit, uint64_t *found) const
{
    std::unique_ptr<Tables> t(new Tables), t2(new Tables);
    std::vector<uint16_t> scratch(kWin + (1u << 20) + 512);
    for (uint32_t w = 0; w < kWin; ++w) scratch[w] = (uint16_t)(kMarker | w);
    BitReader r;
    r.in_end = z + n;
    for (uint64_t bit = from_bit; bit < to_bit; ++bit) {
        // cheap pre-test on the three header bits: BFINAL = 0, BTYPE = 2.  Only dynamic-Huffman blocks qualify: a
        // fixed-Huffman "header" is three bits and any bit string decodes under the fixed code until seven zero bits
        // come along, so fixed blocks cannot be told from noise (real FASTQ streams hardly contain any)
        const uint32_t hdr3 = (uint32_t)((z[bit >> 3] | ((uint32_t)z[(bit >> 3) + 1] << 8)) >> (bit & 7)) & 7u;
        if (hdr3 != 4u) continue;
        {   // second pre-test, still without tables: HLIT <= 29, HDIST <= 29, and the 3-bit code-length code lengths
            // that follow must form a COMPLETE prefix code (every encoder writes one; a random bit string hardly ever does)
            uint64_t w;
            memcpy(&w, z + (bit >> 3), 8);
            w >>= (bit & 7) + 3;
            const uint32_t hlit = (uint32_t)(w & 31u), hdist = (uint32_t)((w >> 5) & 31u), hclen = (uint32_t)((w >> 10) & 15u) + 4;
            if (hlit > 29 || hdist > 29) continue;
            // 14 + 3 * 19 = 71 bits: the lengths may run past this word
            uint64_t lens_bits = w >> 14; // 50 - (bit & 7) - ... valid bits: (64 - (bit&7) - 3 - 14) >= 40 -> 13 lengths
            uint32_t kraft = 0, have = (uint32_t)((64 - (bit & 7) - 17) / 3);
            uint64_t w2 = 0;
            if (hclen > have) {
                memcpy(&w2, z + (bit >> 3) + 8, 8);
                const uint32_t used = 64 - (uint32_t)(bit & 7) - 17; // bits of `lens_bits` that are real
                lens_bits |= used >= 64 ? 0 : (w2 << used);
            }
            bool nonzero = false;
            for (uint32_t i = 0; i < hclen; ++i) {
                const uint32_t l = (uint32_t)((lens_bits >> (3 * i)) & 7u);
                if (l) { kraft += 128u >> l; nonzero = true; }
            }
            if (!nonzero || kraft != 128u) continue;
        }
        r.seek(z, bit);
        bool last = false;
        uint32_t stored = 0, type = 0;
        if (read_block_header(r, *t, &last, &stored, &type)) continue;
        // the block itself
        uint16_t *o = scratch.data() + kWin;
        const char *err = nullptr;
        BlockStatus st;
        size_t produced = 0;
        bool ok = true;
        for (;;) {
            st = huffman_block<uint16_t>(r, *t, o, scratch.data() + scratch.size() - 400, scratch.data(), &err);
            if (st == kBlockError) { ok = false; break; }
            if (st == kBlockEnd) break;
            // block longer than the scratch: keep the last 32 KiB as history and go on
            produced += (size_t)(o - (scratch.data() + kWin));
            memmove(scratch.data(), o - kWin, kWin * sizeof(uint16_t));
            o = scratch.data() + kWin;
            if (produced > (1u << 28)) { ok = false; break; }
        }
        if (!ok) continue;
        if (r.overrun()) continue;
        // a second dynamic header must follow (it may be the final block's)
        BitReader r2 = r;
        bool last2 = false;
        uint32_t stored2 = 0, type2 = 0;
        if (read_block_header(r2, *t2, &last2, &stored2, &type2) || type2 != 2u) continue;
        *found = bit;
        return true;
    }
    return false;
}

void ParallelGunzip::Impl::decode_segment(size_t j)
{
    Segment &s = *segs[j];
    std::unique_ptr<Tables> t(new Tables);
    BitReader r;
    r.in_end = z + n;
    r.seek(z, s.start_bit);
    const size_t comp_bytes = (size_t)(((s.stop_bit == ~0ull ? (uint64_t)n * 8 : s.stop_bit) - s.start_bit) / 8);
    // the first segment knows its (empty) window: it decodes to bytes from the start
    bool symbolic = j != 0;
    s.out_buf = take(pool_out);
    s.out_buf.ensure(std::max<size_t>(comp_bytes * 5, 1u << 20) + 1024);
    if (symbolic) {
        s.sym_buf = take(pool_sym);
        s.sym_buf.ensure(2 * (kWin + std::max<size_t>(comp_bytes * 2, 1u << 20) + 1024));
        for (uint32_t w = 0; w < kWin; ++w) s.sym()[w] = (uint16_t)(kMarker | w);
    }
    size_t so = kWin, bo = 0; // positions (elements) in sym resp. out: the buffers may move when they grow
    auto fail = [&](const char *msg) { s.error = msg; s.failed = true; };
    // A segment's buffers grow with what it inflates to, and on repetitive data the symbolic form never ends (a marker
    // copied at distance 1 stays a marker): 2 GB of 'N' reads in a 12 MB file would put gigabytes into the look-ahead.
    // Beyond this many bytes the multi-threaded decoder declines the file and the sequential one, which streams in
    // constant memory like zlib, takes over (FASTQ text inflates 3-6x).
    const size_t max_ratio = getenv("MHX_PINFLATE_MAX_RATIO") ? (size_t)atol(getenv("MHX_PINFLATE_MAX_RATIO")) : 16;
    const size_t seg_limit = std::max<size_t>(comp_bytes * max_ratio, (size_t)8 << 20);
    static const char *const too_much = "compression ratio beyond the multi-threaded decoder's memory bound";
    for (;;) {
        if (abort.load(std::memory_order_relaxed)) return fail("aborted");
        const uint64_t pos = r.bitpos(z);
        if (pos == s.stop_bit) break;
        if (s.stop_bit != ~0ull && pos > s.stop_bit) return fail("segment ran past the start of its successor");
        if (r.overrun()) return fail("unexpected end of deflate stream");
        bool last = false;
        uint32_t stored = 0, type = 0;
        const char *err = read_block_header(r, *t, &last, &stored, &type);
        if (err) return fail(err);
        if (type == 0) {
            r.unread();
            if (r.in > r.in_end || (size_t)(r.in_end - r.in) < stored) return fail("unexpected end of stored block");
            if ((symbolic ? so - kWin : bo) + stored > seg_limit) return fail(too_much);
            if (symbolic) {
                s.sym_buf.ensure(2 * (so + stored + 1024));
                uint16_t *d = s.sym() + so;
                for (uint32_t i = 0; i < stored; ++i) d[i] = r.in[i];
                so += stored;
            } else {
                s.out_buf.ensure(bo + stored + 1024);
                memcpy(s.out() + bo, r.in, stored);
                bo += stored;
            }
            r.in += stored;
        } else if (symbolic) {
            for (;;) {
                uint16_t *o = s.sym() + so;
                const BlockStatus st = huffman_block<uint16_t>(r, *t, o, s.sym() + s.sym_cap() - 512, s.sym(), &err);
                so = (size_t)(o - s.sym());
                if (st == kBlockError) return fail(err);
                if (st == kBlockEnd) break;
                if (so - kWin > seg_limit) return fail(too_much);
                s.sym_buf.ensure(s.sym_buf.cap * 2);
            }
        } else {
            for (;;) {
                uint8_t *o = s.out() + bo;
                const BlockStatus st = huffman_block<uint8_t>(r, *t, o, s.out() + s.out_buf.cap - 512, s.out(), &err);
                bo = (size_t)(o - s.out());
                if (st == kBlockError) return fail(err);
                if (st == kBlockEnd) break;
                if (bo > seg_limit) return fail(too_much);
                s.out_buf.ensure(s.out_buf.cap * 2);
            }
        }
        if (symbolic) {
            // no marker left in the last 32 KiB: everything from here on is plain bytes
            const size_t ns = so - kWin;
            if (ns >= kWin && !any_marker(s.sym() + so - kWin, kWin)) {
                s.n_sym = ns;
                s.out_buf.ensure(ns + std::max<size_t>(comp_bytes * 5, 1u << 20) + 1024);
                const uint16_t *w = s.sym() + so - kWin;
                uint8_t *d = s.out() + ns - kWin;
                for (uint32_t i = 0; i < kWin; ++i) d[i] = (uint8_t)w[i];
                bo = ns;
                symbolic = false;
            }
        }
        if (last) { s.final_block_seen = true; break; }
    }
    if (symbolic) {
        s.n_sym = so - kWin;
        s.n_out = s.n_sym;
        s.out_buf.ensure(s.n_sym + 64);
    } else {
        s.n_out = bo;
    }
    if (s.stop_bit != ~0ull && !s.final_block_seen && r.bitpos(z) != s.stop_bit) return fail("segment did not end on its successor's block");
    if (s.stop_bit == ~0ull && !s.final_block_seen) return fail("unexpected end of deflate stream");
    r.byte_align();
    s.end_bit = r.bitpos(z);
}

void ParallelGunzip::Impl::worker()
{
    for (;;) {
        const size_t j = next_task.fetch_add(1);
        if (j >= segs.size()) return;
        {   // bounded look-ahead: decoded segments wait in memory until the consumer has taken them
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return abort || (j < consumed + lookahead && (j == consumed || outstanding < budget)); });
            if (abort) return;
        }
        Segment &s = *segs[j];
        const auto t_start = std::chrono::steady_clock::now();
        try {
            decode_segment(j);
        } catch (const std::exception &e) { // bad_alloc, system_error: the segment fails, the process does not
            s.error = e.what();
            s.failed = true;
        }
        {
            std::lock_guard<std::mutex> lk(m);
            s.decoded = true;
            outstanding += 3 * s.n_out;
        }
        cv.notify_all();
        const auto t_decoded = std::chrono::steady_clock::now();
        auto t_waited = t_decoded;
        if (!s.failed && j > 0) {
            // resolution needs the predecessor's last 32 KiB
            Segment &p = *segs[j - 1];
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return abort || p.window_ready || p.failed; });
                if (abort) return;
            }
            t_waited = std::chrono::steady_clock::now();
            if (p.failed) { s.error = p.error; s.failed = true; }
            else try {
                // One table look-up per symbol, no branch: symbols 0..255 map to themselves, marker 0x8000 | w to byte w of
                // the 32 KiB in front of this segment (index kWin - 1 = the byte right before it).  A marker that reaches in
                // front of the stream's first byte (only possible with a corrupt stream) is caught by a sentinel scan below.
                std::unique_ptr<uint8_t[]> lut(new uint8_t[65536]);
                memset(lut.get(), 0, 65536);
                for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i;
                memcpy(lut.get() + kMarker + (kWin - p.window_len), p.window, p.window_len);
                bool bad = false;
                const uint32_t lowest = kWin - p.window_len;
                auto resolve = [&](size_t a, size_t b) {
                    const uint16_t *src = s.sym() + kWin;
                    uint8_t *dst = s.out();
                    const uint8_t *tab = lut.get();
                    if (lowest) { // short history (the very beginning of the stream): check the markers as well
                        for (size_t i = a; i < b; ++i) {
                            const uint16_t v = src[i];
                            if (v >= kMarker && (uint32_t)(v & 0x7FFFu) < lowest) bad = true;
                            dst[i] = tab[v];
                        }
                        return;
                    }
                    size_t i = a;
                    for (; i + 8 <= b; i += 8) {
                        dst[i] = tab[src[i]]; dst[i + 1] = tab[src[i + 1]]; dst[i + 2] = tab[src[i + 2]]; dst[i + 3] = tab[src[i + 3]];
                        dst[i + 4] = tab[src[i + 4]]; dst[i + 5] = tab[src[i + 5]]; dst[i + 6] = tab[src[i + 6]]; dst[i + 7] = tab[src[i + 7]];
                    }
                    for (; i < b; ++i) dst[i] = tab[src[i]];
                };
                // own window first, so that the successor can start resolving
                const size_t tail_from = s.n_out > kWin ? s.n_out - kWin : 0;
                if (tail_from < s.n_sym) resolve(tail_from, s.n_sym);
                {
                    std::lock_guard<std::mutex> lk(m);
                    s.window_len = (uint32_t)std::min<size_t>(kWin, s.n_out + p.window_len);
                    if (s.n_out >= kWin) memcpy(s.window, s.out() + s.n_out - kWin, kWin);
                    else { // shorter than a window: the predecessor's tail comes first
                        const size_t keep = s.window_len - s.n_out;
                        memcpy(s.window, p.window + (p.window_len - keep), keep);
                        memcpy(s.window + keep, s.out(), s.n_out);
                    }
                    s.window_ready = true;
                }
                cv.notify_all();
                resolve(0, std::min(tail_from, s.n_sym));
                if (bad) { s.error = "invalid distance too far back"; s.failed = true; }
                give(pool_sym, std::move(s.sym_buf));
            } catch (const std::exception &e) { // the 64 KiB table
                s.error = e.what();
                s.failed = true;
            }
        } else if (!s.failed) {
            std::lock_guard<std::mutex> lk(m);
            s.window_len = (uint32_t)std::min<size_t>(kWin, s.n_out);
            memcpy(s.window, s.out() + s.n_out - s.window_len, s.window_len);
            s.window_ready = true;
        }
        const auto t_resolved = std::chrono::steady_clock::now();
        if (!s.failed) s.crc = (uint32_t)crc32_update(0, s.out(), s.n_out);
        const auto t_crc = std::chrono::steady_clock::now();
        {
            std::lock_guard<std::mutex> lk(m);
            s.done = true;
            if (s.failed) s.window_ready = false;
        }
        cv.notify_all();
        if (getenv("MHX_PINFLATE_DEBUG"))
        {
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "segment %zu: %zu bytes, %zu symbolic (%.0f%%)%s  decode %.1f ms  wait %.1f  resolve %.1f  crc %.1f\n", j, s.n_out, s.n_sym,
                    s.n_out ? 100.0 * s.n_sym / s.n_out : 0.0, s.failed ? " FAILED" : "", ms(t_start, t_decoded), ms(t_decoded, t_waited), ms(t_waited, t_resolved), ms(t_resolved, t_crc));
        }
    }
}

ParallelGunzip::ParallelGunzip() : impl_(new Impl) {}
ParallelGunzip::~ParallelGunzip() { delete impl_; }
const std::string &ParallelGunzip::error() const { return impl_->error; }
size_t ParallelGunzip::consumed_input() const { return impl_->trailer_off ? impl_->trailer_off + 8 : 0; }

bool ParallelGunzip::start(const uint8_t *z, size_t n, int threads, size_t min_bytes_arg, size_t seg_bytes_arg)
{
    Impl &p = *impl_;
    p.z = z;
    p.n = n;
    p.nthreads = threads;
    p.deflate_off = member_header_len(z, n);
    // small inputs: the sequential decoder is as fast (MHX_PINFLATE_MIN / MHX_PINFLATE_SEGMENT: test knobs)
    const size_t min_env = getenv("MHX_PINFLATE_MIN") ? (size_t)atol(getenv("MHX_PINFLATE_MIN")) : 0;
    const size_t min_bytes = min_env ? min_env : (min_bytes_arg ? min_bytes_arg : (8u << 20));
    if (!p.deflate_off || threads < 2 || n < min_bytes) return false;
    // segment targets: equal shares of the compressed bytes, ~1 MiB each (small segments keep the working set of a worker
    // -- 16-bit symbols plus bytes of ~10 MB of output -- near the caches and let the buffer pool recycle early), at least
    // two per thread
    const size_t seg_env = getenv("MHX_PINFLATE_SEGMENT") ? (size_t)atol(getenv("MHX_PINFLATE_SEGMENT")) : 0;
    const size_t seg_bytes = seg_env ? seg_env : (seg_bytes_arg ? seg_bytes_arg : (1u << 20));
    size_t nseg = std::max<size_t>((size_t)threads * 2, n / seg_bytes);
    if (nseg > 65536) nseg = 65536;
    const uint64_t first_bit = (uint64_t)p.deflate_off * 8, end_bit = (uint64_t)n * 8;
    const uint64_t share = (end_bit - first_bit) / nseg;
    std::vector<uint64_t> starts(nseg, ~0ull);
    starts[0] = first_bit;
    {   // block search for every target, all threads
        std::atomic<size_t> next{1};
        std::vector<std::thread> ts;
        auto search = [&] {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= nseg) return;
                const uint64_t from = first_bit + share * j, to = std::min(end_bit - 64, from + share);
                uint64_t found;
                try {
                    if (from < to && p.find_block_start(from, to, &found)) starts[j] = found;
                } catch (const std::exception &) { // out of memory in a trial decode: no start found in this share
                }
            }
        };
        try {
            for (int t = 0; t < threads; ++t) ts.emplace_back(search);
        } catch (const std::system_error &) { // the host would not give us another thread: the ones we have do the work
        }
        if (ts.empty()) search();
        for (auto &t : ts) t.join();
    }
    for (size_t j = 0; j < nseg; ++j) {
        if (starts[j] == ~0ull) continue; // no block starts inside this share (a very long block): merged into the previous one
        std::unique_ptr<Segment> s(new Segment);
        s->start_bit = starts[j];
        if (!p.segs.empty()) p.segs.back()->stop_bit = starts[j];
        s->stop_bit = ~0ull;
        p.segs.push_back(std::move(s));
    }
    if (p.segs.size() < 2) { p.segs.clear(); return false; }
    p.lookahead = (size_t)threads * 2;
    if (const char *e = getenv("MHX_PINFLATE_BUDGET_MB")) p.budget = (size_t)atol(e) << 20;
    try {
        for (int t = 0; t < threads; ++t) p.workers.emplace_back([&p] { p.worker(); });
    } catch (const std::system_error &) {
        if (p.workers.empty()) { p.segs.clear(); return false; } // no thread at all: the sequential decoder
    }
    return true;
}

// Copies up to `want` bytes of the member's output, in order.  Returns the number of bytes, 0 at the end of the
// member (then consumed_input() tells where the next member would start) or (size_t)-1 on error.
size_t ParallelGunzip::read(uint8_t *dst, size_t want)
{
    Impl &p = *impl_;
    if (p.failed) return (size_t)-1;
    size_t got = 0;
    while (got < want && !p.finished) {
        Segment &s = *p.segs[p.cur];
        {
            std::unique_lock<std::mutex> lk(p.m);
            p.cv.wait(lk, [&] { return s.done; });
        }
        if (s.failed) { p.failed = true; p.error = s.error; return (size_t)-1; }
        if (p.cur_off == 0) { // first touch of this segment: account for it
            p.crc_all = p.cur == 0 ? s.crc : (uint32_t)crc32_combine(p.crc_all, s.crc, (z_off_t)s.n_out);
            p.total_out += s.n_out;
        }
        const size_t take = std::min(want - got, s.n_out - p.cur_off);
        p.copier.copy(dst + got, s.out() + p.cur_off, take);
        got += take;
        p.cur_off += take;
        if (p.cur_off == s.n_out) {
            // the member's final block ends the run, wherever it is (segments cut further on belong to later members)
            const bool last = s.final_block_seen || p.cur + 1 == p.segs.size();
            if (last) {
                // trailer: CRC-32 and ISIZE of the whole member
                const size_t off = (size_t)(s.end_bit / 8);
                if (!s.final_block_seen || p.n - off < 8) { p.failed = true; p.error = "unexpected end of file"; return (size_t)-1; }
                const uint8_t *t = p.z + off;
                const uint32_t want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
                const uint32_t want_len = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
                if (want_len != (uint32_t)p.total_out) { p.failed = true; p.error = "incorrect length check"; return (size_t)-1; }
                if (want_crc != p.crc_all) { p.failed = true; p.error = "incorrect data check"; return (size_t)-1; }
                p.trailer_off = off;
                p.finished = true;
            }
            p.give(p.pool_out, std::move(s.out_buf));
            {
                std::lock_guard<std::mutex> lk(p.m);
                p.consumed = p.cur + 1;
                p.outstanding -= std::min(p.outstanding, 3 * s.n_out);
            }
            p.cv.notify_all();
            ++p.cur;
            p.cur_off = 0;
        }
    }
    return got;
}

// ---- BGZF -----------------------------------------------------------------------------------------------------------
// Every block is a gzip member of its own (no match reaches into another block) whose header carries the block's
// compressed size (extra subfield 'B','C') and whose trailer its CRC-32 and length: the chain of blocks is walked
// without decoding anything, which gives every block its place in the output, and the blocks are then decoded by the
// worker threads with the ordinary byte decoder, each into a scratch buffer (the decoder may write a few bytes past a
// block's end) and from there into its group's buffer.  Groups of kGroup blocks (<= 32 MiB) are the unit handed to the
// consumer; the workers run at most kSlots groups ahead of it.
struct BgzfReader::Impl {
    struct Block { size_t off; uint32_t csize, isize; uint32_t out_off; }; // out_off: inside the block's group
    static constexpr size_t kGroup = 512, kSlots = 3;
    const uint8_t *z = nullptr;
    size_t n = 0, end_off = 0;
    std::vector<Block> blocks;
    std::vector<uint32_t> group_bytes;         // output bytes of group g
    std::vector<std::vector<uint8_t>> slot;    // kSlots group buffers
    std::vector<uint32_t> left;                // blocks of group g still being decoded
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<size_t> next_block{0};
    size_t consumed_groups = 0;                // groups the consumer is done with
    CopyTeam copier;                           // read(): the consumer's copies
    bool abort = false, failed = false;
    std::string error;
    size_t cur_group = 0, cur_off = 0;         // consumer position

    // size of the BGZF block at z[off..), 0 if there is none
    static uint32_t block_size(const uint8_t *z, size_t n, size_t off)
    {
        if (n - off < 28 || z[off] != 0x1f || z[off + 1] != 0x8b || z[off + 2] != 8 || !(z[off + 3] & 4)) return 0;
        const uint32_t xlen = z[off + 10] | (z[off + 11] << 8);
        if (n - off < 12 + (size_t)xlen) return 0;
        for (uint32_t p = 0; p + 4 <= xlen;) {
            const uint8_t *f = z + off + 12 + p;
            const uint32_t slen = f[2] | (f[3] << 8);
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && p + 6 <= xlen) {
                const uint32_t total = (uint32_t)(f[4] | (f[5] << 8)) + 1u;
                return total >= 12 + xlen + 8 && total <= n - off ? total : 0;
            }
            p += 4 + slen;
        }
        return 0;
    }
    void worker()
    {
        GzInflater inf;
        std::vector<uint8_t> scratch(65536 + GzInflater::kOvershoot + 64);
        for (;;) {
            const size_t i = next_block.fetch_add(1);
            if (i >= blocks.size()) return;
            const size_t g = i / kGroup;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return abort || g < consumed_groups + kSlots; });
                if (abort) return;
            }
            const Block &b = blocks[i];
            bool ok = true;
            { // (also a block that announces no data, e.g. the end-of-file marker: its CRC and length are checked like any other's)
                inf.set_input(z + b.off, b.csize); // one member: header, deflate stream, CRC-32 and length (both checked)
                size_t got = 0;
                // room for one byte more than the block announces: the decoder then runs through the end-of-block symbol
                // and the trailer (CRC-32, length) by itself; a stream that produces that byte is refused
                while (ok && !inf.done()) {
                    const size_t r = inf.inflate(scratch.data() + got, (size_t)b.isize + 1 - got, scratch.data());
                    if (r == (size_t)-1 || (r == 0 && !inf.done())) ok = false;
                    else got += r;
                    if (got > b.isize) ok = false;
                }
                if (ok && got != b.isize) ok = false;
                if (ok) memcpy(slot[g % kSlots].data() + b.out_off, scratch.data(), b.isize);
            }
            std::lock_guard<std::mutex> lk(m);
            if (!ok && !failed) { failed = true; error = inf.error().empty() ? "corrupt BGZF block" : inf.error(); }
            if (--left[g] == 0 || !ok) cv.notify_all();
        }
    }
    ~Impl()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            abort = true;
        }
        cv.notify_all();
        for (auto &t : workers) t.join();
    }
};

BgzfReader::BgzfReader() : impl_(new Impl) {}
BgzfReader::~BgzfReader() { delete impl_; }
const std::string &BgzfReader::error() const { return impl_->error; }
size_t BgzfReader::consumed_input() const { return impl_->end_off; }

bool BgzfReader::start(const uint8_t *z, size_t n, int threads)
{
    Impl &p = *impl_;
    if (threads < 2 || n < 28) return false;
    size_t off = 0;
    uint32_t in_group = 0;
    while (off < n) {
        const uint32_t len = Impl::block_size(z, n, off);
        if (!len) break;
        const uint8_t *t = z + off + len - 4;
        const uint32_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        if (isize > 65536) break; // not BGZF after all
        if (p.blocks.size() % Impl::kGroup == 0) { p.group_bytes.push_back(0); in_group = 0; }
        p.blocks.push_back({off, len, isize, in_group});
        in_group += isize;
        p.group_bytes.back() = in_group;
        off += len;
    }
    if (p.blocks.size() < 8) { p.blocks.clear(); p.group_bytes.clear(); return false; } // a handful of blocks: not worth the threads
    p.z = z;
    p.n = n;
    p.end_off = off;
    const size_t ngroups = p.group_bytes.size();
    p.left.resize(ngroups);
    for (size_t g = 0; g < ngroups; ++g) p.left[g] = (uint32_t)std::min(Impl::kGroup, p.blocks.size() - g * Impl::kGroup);
    p.slot.resize(Impl::kSlots);
    for (auto &s : p.slot) s.resize(Impl::kGroup * 65536);
    try {
        for (int t = 0; t < threads; ++t) p.workers.emplace_back([&p] { p.worker(); });
    } catch (const std::system_error &) { // the host would not give us another thread: the ones we have do the work
        if (p.workers.empty()) { p.blocks.clear(); p.group_bytes.clear(); return false; } // none at all: the sequential decoder
    }
    return true;
}

size_t BgzfReader::read(uint8_t *dst, size_t want)
{
    Impl &p = *impl_;
    size_t got = 0;
    while (got < want && p.cur_group < p.group_bytes.size()) {
        const size_t g = p.cur_group;
        {
            std::unique_lock<std::mutex> lk(p.m);
            p.cv.wait(lk, [&] { return p.failed || p.left[g] == 0; });
            if (p.failed) return (size_t)-1;
        }
        const size_t take = std::min<size_t>(want - got, p.group_bytes[g] - p.cur_off);
        p.copier.copy(dst + got, p.slot[g % Impl::kSlots].data() + p.cur_off, take);
        got += take;
        p.cur_off += take;
        if (p.cur_off == p.group_bytes[g]) {
            {
                std::lock_guard<std::mutex> lk(p.m);
                p.consumed_groups = g + 1;
            }
            p.cv.notify_all();
            ++p.cur_group;
            p.cur_off = 0;
        }
    }
    return got;
}

} // namespace mhx

// ---- C ABI: whole-buffer gunzip with several threads (tests, and callers that hold the compressed bytes) ----------
using namespace mhx;

extern "C" int mhx_gunzip_buffer_mt(const void *gz, size_t n, void *out, size_t cap, size_t *out_n, int threads)
{
    clear_error();
    if (!gz || !out_n) return fail(MHX_E_ARG, "null argument");
    try {
        std::vector<uint8_t> in(n + GzInflater::kInputPad, 0);
        memcpy(in.data(), gz, n);
        uint8_t *dst = (uint8_t *)out;
        size_t total = 0, off = 0;
        BgzfReader bgzf;
        if (threads >= 2 && bgzf.start(in.data(), n, threads)) { // bgzip output: independent blocks, decoded side by side
            std::vector<uint8_t> piece(4u << 20);
            for (;;) {
                const size_t room = dst && total < cap ? cap - total : 0;
                const size_t got = room >= piece.size() ? bgzf.read(dst + total, room) : bgzf.read(piece.data(), piece.size());
                if (got == (size_t)-1) return fail(MHX_E_FORMAT, "gunzip: %s", bgzf.error().c_str());
                if (got == 0) break;
                if (room < piece.size() && dst && total < cap) memcpy(dst + total, piece.data(), std::min(got, cap - total));
                total += got;
            }
            off = bgzf.consumed_input();
        }
        bool declined = false;
        // every large member in turn (`cat a.gz b.gz`): the first one that the multi-threaded decoder does not take, and
        // everything behind it, goes to the sequential decoder below
        while (off < n && threads >= 2 && !declined) {
            ParallelGunzip par;
            if (!par.start(in.data() + off, n - off, threads)) break;
            const size_t total_before = total;
            std::vector<uint8_t> piece(4u << 20);
            for (;;) {
                const size_t room = dst && total < cap ? cap - total : 0;
                const size_t got = room >= piece.size() ? par.read(dst + total, room) : par.read(piece.data(), piece.size());
                if (got == (size_t)-1) {
                    if (par.error().find("memory bound") != std::string::npos) { total = total_before; declined = true; break; } // the sequential decoder streams it
                    return fail(MHX_E_FORMAT, "gunzip: %s", par.error().c_str());
                }
                if (got == 0) break;
                if (room < piece.size() && dst && total < cap) memcpy(dst + total, piece.data(), std::min(got, cap - total));
                total += got;
            }
            if (!declined) off += par.consumed_input();
        }
        // further members (or everything, when the parallel decoder declined): the sequential decoder
        if (off < n) {
            size_t rest = 0;
            const int rc = mhx_gunzip_buffer(in.data() + off, n - off, dst && total < cap ? dst + total : nullptr, dst && total < cap ? cap - total : 0, &rest);
            if (rc && rc != MHX_E_CAPACITY) return rc;
            total += rest;
        }
        *out_n = total;
        if (dst && total > cap) return fail(MHX_E_CAPACITY, "gunzip: output buffer too small (%zu needed)", total);
        return MHX_OK;
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_gunzip_buffer_mt: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_gunzip_buffer_mt: %s", e.what());
    }
}
