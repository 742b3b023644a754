// mhx_inflate.cpp -- gzip (RFC 1952) / DEFLATE (RFC 1951) decoder of the FASTQ ingest.
//
// AuriClass hands `.fq.gz` paths straight to `mash sketch` (/root/reference/auriclass/classes.py:588),
// which inflates them with zlib inside kseq; once the sketch runs on the GPU the inflate is what a
// sample waits for, so this is a decoder built for throughput rather than zlib's generality:
// the whole compressed file is in memory, the bit buffer is 64 bits wide and refilled with one
// unaligned load, literal/length and distance codes are decoded through 11- and 8-bit first-level
// tables with sub-tables for longer codes, matches are copied eight bytes at a time, and output goes
// straight into the caller's chunk buffer (which carries the previous 32 KiB in front of it), in
// resumable pieces.  Written from the two RFCs; checked against zlib on every block type in
// tests/test_lib_cpu.py.
#include <stdlib.h>
#include <string.h>
#include <mutex>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "mhx_inflate_impl.h"
#include "mhx_internal.h"

namespace mhx {

using namespace deflate;

struct GzInflater::Impl {
    BitReader r;
    Tables t;
    enum State { kMemberHeader, kBlockHeader, kStored, kHuffman, kTrailer, kDone, kError } state = kMemberHeader;
    bool last_block = false;
    uint32_t stored_left = 0;
    uint32_t crc = 0;
    uint64_t member_out = 0;
    std::string error;
    bool verify_crc = true;
    // deferred checking: inflate() returns at every member end and reports the trailer's CRC instead of checking it
    bool defer_crc = false, member_end_pending = false;
    uint32_t pending_crc = 0;

    bool fail(const char *msg) { error = msg; state = kError; return false; }
    bool read_member_header();
    bool read_block_header();
    bool read_trailer();
};

bool GzInflater::Impl::read_member_header()
{
    // between members: skip nothing; no more input -> done.  Trailing bytes that are not a gzip
    // header are ignored, as gzread does.
    r.byte_align();
    r.unread(); // whole unread bytes go back so that r.in is the true position
    const uint8_t *in = r.in, *const in_end = r.in_end;
    if (in_end - in < 18) { state = kDone; return true; }
    if (in[0] != 0x1f || in[1] != 0x8b) { state = kDone; return true; }
    if (in[2] != 8) return fail("unsupported gzip compression method");
    const uint8_t flg = in[3];
    const uint8_t *p = in + 10;
    if (flg & 4) { // FEXTRA
        if (in_end - p < 2) return fail("truncated gzip header");
        const size_t xlen = p[0] | (p[1] << 8);
        p += 2;
        if ((size_t)(in_end - p) < xlen) return fail("truncated gzip header");
        p += xlen;
    }
    for (int bit = 8; bit <= 16; bit <<= 1) { // FNAME, FCOMMENT: zero-terminated
        if (!(flg & bit)) continue;
        const void *z = memchr(p, 0, (size_t)(in_end - p));
        if (!z) return fail("truncated gzip header");
        p = (const uint8_t *)z + 1;
    }
    if (flg & 2) { if (in_end - p < 2) return fail("truncated gzip header"); p += 2; } // FHCRC
    r.in = p;
    crc = 0;
    member_out = 0;
    state = kBlockHeader;
    return true;
}

bool GzInflater::Impl::read_block_header()
{
    uint32_t type = 0, len = 0;
    const char *err = deflate::read_block_header(r, t, &last_block, &len, &type);
    if (err) return fail(err);
    if (type == 0) { stored_left = len; state = kStored; }
    else state = kHuffman;
    return true;
}

bool GzInflater::Impl::read_trailer()
{
    r.byte_align();
    r.unread();
    const uint8_t *in = r.in;
    if (r.in_end - in < 8) return fail("unexpected end of file");
    const uint32_t want_crc = in[0] | (in[1] << 8) | (in[2] << 16) | ((uint32_t)in[3] << 24);
    const uint32_t want_len = in[4] | (in[5] << 8) | (in[6] << 16) | ((uint32_t)in[7] << 24);
    r.in += 8;
    if (want_len != (uint32_t)member_out) return fail("incorrect length check");
    if (defer_crc) { member_end_pending = true; pending_crc = want_crc; }
    else if (verify_crc && want_crc != crc) return fail("incorrect data check");
    state = kMemberHeader;
    return true;
}

// ---- CRC-32 (IEEE 802.3 polynomial, as gzip), slicing by 8 -----------------------------------
static uint32_t g_crc_table[8][256];
static void build_crc_tables()
{
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xFF];
}
static void init_crc()
{ // the inflate threads of a paired sample and their CRC followers all come through here
    static std::once_flag once;
    std::call_once(once, build_crc_tables);
}
static uint32_t crc32_tables(uint32_t c, const uint8_t *p, size_t n) // c and result are the raw (inverted) register
{
    while (n && ((uintptr_t)p & 7)) { c = (c >> 8) ^ g_crc_table[0][(c ^ *p++) & 0xFF]; --n; }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = g_crc_table[7][v & 0xFF] ^ g_crc_table[6][(v >> 8) & 0xFF] ^ g_crc_table[5][(v >> 16) & 0xFF] ^ g_crc_table[4][(v >> 24) & 0xFF] ^
            g_crc_table[3][(v >> 32) & 0xFF] ^ g_crc_table[2][(v >> 40) & 0xFF] ^ g_crc_table[1][(v >> 48) & 0xFF] ^ g_crc_table[0][v >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ g_crc_table[0][(c ^ *p++) & 0xFF];
    return c;
}

#if defined(__x86_64__)
// Carry-less-multiply folding (Gopal et al., "Fast CRC Computation for Generic Polynomials Using
// PCLMULQDQ"): four 128-bit lanes are folded across 64 bytes per step, then reduced to 32 bits with a
// Barrett step.  Constants are x^n mod P for the bit-reflected gzip polynomial.  n >= 64, n % 16 == 0.
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t c, const uint8_t *p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    p += 64;
    n -= 64;
    while (n >= 64) {
        const __m128i y1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), y2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i y3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), y4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, y1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, y2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, y3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, y4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64;
        n -= 64;
    }
    __m128i y = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), y), x2);
    y = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), y), x3);
    y = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), y), x4);
    while (n >= 16) {
        y = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), y), _mm_loadu_si128((const __m128i *)p));
        p += 16;
        n -= 16;
    }
    // 128 -> 64 -> 32 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5k0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10);
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x2, mask32), poly, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
static bool have_clmul()
{
    static const bool yes = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return yes;
}
#endif

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    init_crc();
    uint32_t c = ~crc;
#if defined(__x86_64__)
    static const bool tables_only = getenv("MHX_CRC_TABLES") != nullptr; // test knob: force the portable path
    if (n >= 256 && have_clmul() && !tables_only) {
        const size_t bulk = n & ~(size_t)15;
        c = crc32_clmul(c, p, bulk);
        p += bulk;
        n -= bulk;
    }
#endif
    return ~crc32_tables(c, p, n);
}

// ---- public wrapper -----------------------------------------------------------------------------
GzInflater::GzInflater() : impl_(new Impl()) { init_crc(); }
GzInflater::~GzInflater() { delete impl_; }
void GzInflater::set_verify_crc(bool on) { impl_->verify_crc = on; }
void GzInflater::set_deferred_crc(bool on) { impl_->defer_crc = on; }
bool GzInflater::take_member_end(uint32_t *crc)
{
    if (!impl_->member_end_pending) return false;
    impl_->member_end_pending = false;
    *crc = impl_->pending_crc;
    return true;
}
const std::string &GzInflater::error() const { return impl_->error; }
bool GzInflater::done() const { return impl_->state == Impl::kDone; }

void GzInflater::set_input(const uint8_t *data, size_t n)
{
    Impl &s = *impl_;
    s.r.in = data;
    s.r.in_end = data + n;
    s.r.bitbuf = 0;
    s.r.bitcnt = 0;
    s.state = Impl::kMemberHeader;
    s.error.clear();
}

// Decodes until `out + limit` is reached (may overshoot by up to kOvershoot bytes), the input ends or an
// error occurs.  `out` must be preceded by the previous 32 KiB of output (or be the very start of the
// stream) and have kOvershoot + 8 writable bytes past `limit`.  Returns the number of bytes produced, or
// (size_t)-1 on error.
size_t GzInflater::inflate(uint8_t *out, size_t limit, const uint8_t *window_start)
{
    Impl &s = *impl_;
    uint8_t *o = out;
    uint8_t *const o_limit = out + limit;
    uint8_t *crc_from = out;
    auto account = [&]() {
        const size_t n = (size_t)(o - crc_from);
        if (n) {
            if (s.verify_crc && !s.defer_crc) s.crc = crc32_update(s.crc, crc_from, n);
            s.member_out += n;
            crc_from = o;
        }
    };
    for (;;) {
        switch (s.state) {
        case Impl::kMemberHeader:
            if (!s.read_member_header()) return (size_t)-1;
            break;
        case Impl::kBlockHeader:
            if (s.r.overrun()) { s.fail("unexpected end of deflate stream"); return (size_t)-1; }
            if (!s.read_block_header()) return (size_t)-1;
            break;
        case Impl::kStored: {
            // byte aligned: hand back buffered whole bytes, then copy
            s.r.unread();
            size_t n = s.stored_left;
            if (s.r.in > s.r.in_end || (size_t)(s.r.in_end - s.r.in) < n) { s.fail("unexpected end of stored block"); return (size_t)-1; }
            if (o + n > o_limit) n = o < o_limit ? (size_t)(o_limit - o) : 0;
            memcpy(o, s.r.in, n);
            o += n;
            s.r.in += n;
            s.stored_left -= (uint32_t)n;
            if (s.stored_left == 0) s.state = s.last_block ? Impl::kTrailer : Impl::kBlockHeader;
            if (o >= o_limit && s.stored_left) { account(); return (size_t)(o - out); }
            break;
        }
        case Impl::kHuffman: {
            const char *err = nullptr;
            const BlockStatus st = huffman_block<uint8_t>(s.r, s.t, o, o_limit, window_start, &err);
            if (st == kBlockError) { s.fail(err); return (size_t)-1; }
            if (st == kBlockEnd) s.state = s.last_block ? Impl::kTrailer : Impl::kBlockHeader;
            else { account(); return (size_t)(o - out); } // output limit reached inside the block
            break;
        }
        case Impl::kTrailer:
            account();
            if (!s.read_trailer()) return (size_t)-1;
            if (s.member_end_pending) return (size_t)(o - out); // the caller closes the member's CRC here
            break;
        case Impl::kDone:
            account();
            return (size_t)(o - out);
        case Impl::kError:
            return (size_t)-1;
        }
        if (o >= o_limit && s.state != Impl::kTrailer && s.state != Impl::kMemberHeader) { account(); return (size_t)(o - out); }
    }
}

} // namespace mhx

// ---- C ABI: whole-buffer inflate (what the ingest threads run, exposed for tests and callers that
// already hold the compressed bytes) ------------------------------------------------------------------
using namespace mhx;

extern "C" int mhx_gunzip_buffer(const void *gz, size_t n, void *out, size_t cap, size_t *out_n)
{
    clear_error();
    if (!gz || !out_n) return fail(MHX_E_ARG, "null argument");
    // private padded copy of the input (the decoder's 8-byte refills run past the end)
    std::vector<uint8_t> in(n + GzInflater::kInputPad, 0);
    memcpy(in.data(), gz, n);
    GzInflater inf;
    inf.set_input(in.data(), n);
    // decode in pieces into a scratch buffer with the 32 KiB history in front, counting first
    constexpr size_t kPiece = 1u << 20;
    std::vector<uint8_t> buf(GzInflater::kWindow + kPiece + GzInflater::kOvershoot + 16);
    size_t total = 0, hist = 0;
    uint8_t *dst = (uint8_t *)out;
    for (;;) {
        uint8_t *o = buf.data() + GzInflater::kWindow;
        const size_t got = inf.inflate(o, kPiece, o - hist);
        if (got == (size_t)-1) return fail(MHX_E_FORMAT, "gunzip: %s", inf.error().c_str());
        if (dst && total + got <= cap) memcpy(dst + total, o, got);
        total += got;
        if (inf.done()) break;
        // slide: keep the last 32 KiB in front of the next piece
        const size_t have = hist + got;
        const size_t keep = have < GzInflater::kWindow ? have : GzInflater::kWindow;
        memmove(buf.data() + GzInflater::kWindow - keep, o + got - keep, keep);
        hist = keep;
    }
    *out_n = total;
    if (dst && total > cap) return fail(MHX_E_CAPACITY, "gunzip: output buffer too small (%zu needed)", total);
    return MHX_OK;
}
