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

#include "mhx_internal.h"

namespace mhx {

namespace {

constexpr int kLitBits = 11, kDistBits = 8;
constexpr int kValShift = 17, kExtraShift = 13;

constexpr uint32_t kKindLiteral = 0x0100, kKindEnd = 0x0200, kKindSub = 0x0400, kKindInvalid = 0x0800, kKindBase = 0x1000;

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t kClenOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

static inline uint32_t reverse_bits(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

// Builds a two-level decode table for a canonical Huffman code.  payload(sym) gives bits 8..31 of
// the entries of symbol `sym`.  Returns false for an over-subscribed code; incomplete codes are
// legal (unused slots decode as invalid).
template <class Payload>
static bool build_table(const uint8_t *lens, int nsym, int first_bits, uint32_t *table, int table_cap, Payload payload)
{
    int count[16] = {0};
    for (int i = 0; i < nsym; ++i) ++count[lens[i]];
    count[0] = 0;
    int max_len = 15;
    while (max_len > 0 && count[max_len] == 0) --max_len;
    uint32_t next_code[17];
    uint32_t code = 0;
    int64_t left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
        code = (code + (uint32_t)count[l - 1]) << 1;
        next_code[l] = code;
    }
    const int first_size = 1 << first_bits;
    for (int i = 0; i < first_size; ++i) table[i] = kKindInvalid | 1u; // consume one bit, report invalid
    int sub_next = first_size;
    const int sub_bits_max = max_len > first_bits ? max_len - first_bits : 0;
    for (int sym = 0; sym < nsym; ++sym) {
        const int l = lens[sym];
        if (!l) continue;
        const uint32_t c = next_code[l]++;
        const uint32_t r = reverse_bits(c, l);
        if (l <= first_bits) {
            const uint32_t e = payload(sym) | (uint32_t)l;
            for (uint32_t i = r; i < (uint32_t)first_size; i += 1u << l) table[i] = e;
        } else {
            const uint32_t lo = r & (uint32_t)(first_size - 1);
            uint32_t head = table[lo];
            if (!(head & kKindSub)) { // open a sub-table for this prefix
                if (sub_next + (1 << sub_bits_max) > table_cap) return false;
                head = kKindSub | (uint32_t)sub_bits_max | ((uint32_t)sub_next << kValShift);
                table[lo] = head;
                for (int i = 0; i < (1 << sub_bits_max); ++i) table[sub_next + i] = kKindInvalid | 1u;
                sub_next += 1 << sub_bits_max;
            }
            const uint32_t base = head >> kValShift;
            const uint32_t e = payload(sym) | (uint32_t)(l - first_bits);
            for (uint32_t i = r >> first_bits; i < (1u << sub_bits_max); i += 1u << (l - first_bits)) table[base + i] = e;
        }
    }
    return true;
}

} // namespace

// Table entries: bits 0..7 = bits to consume (or index bits of the sub-table), bit 8 literal, bit 9
// end of block, bit 10 sub-table link, bit 11 invalid, bit 12 length / distance base; bits 13..16 = number
// of extra bits that follow the code; bits 17..31 = the literal, the base length, the base distance or the
// sub-table offset.
struct GzInflater::Impl {
    const uint8_t *in = nullptr, *in_end = nullptr; // in_end excludes the kInputPad readable pad bytes
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    enum State { kMemberHeader, kBlockHeader, kStored, kHuffman, kTrailer, kDone, kError } state = kMemberHeader;
    bool last_block = false;
    uint32_t stored_left = 0;
    uint32_t lit[(1 << kLitBits) + 288 * 16];
    uint32_t dist[(1 << kDistBits) + 32 * 128];
    uint32_t crc = 0;
    uint64_t member_out = 0;
    std::string error;
    bool verify_crc = true;
    // deferred checking: inflate() returns at every member end and reports the trailer's CRC instead of checking it
    bool defer_crc = false, member_end_pending = false;
    uint32_t pending_crc = 0;

    // Over-read discipline: the true read position is P = in - (bitcnt >> 3); a refill loads 8 bytes at
    // `in` <= P + 7, i.e. touches bytes up to P + 14.  Every refill below is preceded (at a distance of at
    // most 2 consumed bytes) by an input_overrun() test that pins P <= in_end, so no load reaches past
    // in_end + 17 -- inside the kInputPad (64) zero bytes the caller guarantees behind the input.
    void refill()
    {
        // branch-free refill: valid while 8 bytes at `in` are readable (the buffer is padded)
        uint64_t v;
        memcpy(&v, in, 8);
        bitbuf |= v << bitcnt;
        in += (63 - bitcnt) >> 3;
        bitcnt |= 56;
    }
    uint32_t peek(int n) const { return (uint32_t)(bitbuf & ((1ull << n) - 1)); }
    void drop(int n) { bitbuf >>= n; bitcnt -= n; }
    uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
    bool input_overrun() const { return (in - ((bitcnt) >> 3)) > in_end; }
    void byte_align() { drop(bitcnt & 7); }
    bool fail(const char *msg) { error = msg; state = kError; return false; }

    bool read_member_header();
    bool read_block_header();
    bool read_trailer();
};

bool GzInflater::Impl::read_member_header()
{
    // between members: skip nothing; no more input -> done.  Trailing bytes that are not a gzip
    // header are ignored, as gzread does.
    byte_align();
    // give back whole unread bytes so that `in` is the true position
    in -= bitcnt >> 3;
    bitbuf = 0;
    bitcnt = 0;
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
    in = p;
    crc = 0;
    member_out = 0;
    state = kBlockHeader;
    return true;
}

bool GzInflater::Impl::read_block_header()
{
    refill();
    last_block = take(1) != 0;
    const uint32_t type = take(2);
    if (type == 0) {
        byte_align();
        if (input_overrun()) return fail("unexpected end of deflate stream");
        refill();
        const uint32_t len = take(16), nlen = take(16);
        // LEN/NLEN must lie inside the input: read from the zero pad they would pass the check as 0xFFFF/0x0000
        if (input_overrun()) return fail("unexpected end of deflate stream");
        if ((len ^ nlen) != 0xFFFFu) return fail("stored block length check failed");
        stored_left = len;
        state = kStored;
        return true;
    }
    uint8_t lens[288 + 32];
    int nlit, ndist;
    if (type == 1) {
        nlit = 288; ndist = 32;
        for (int i = 0; i < 144; ++i) lens[i] = 8;
        for (int i = 144; i < 256; ++i) lens[i] = 9;
        for (int i = 256; i < 280; ++i) lens[i] = 7;
        for (int i = 280; i < 288; ++i) lens[i] = 8;
        for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
    } else if (type == 2) {
        nlit = (int)take(5) + 257;
        ndist = (int)take(5) + 1;
        const int nclen = (int)take(4) + 4;
        if (nlit > 286 || ndist > 30) return fail("too many length or distance symbols");
        uint8_t clens[19] = {0};
        if (input_overrun()) return fail("unexpected end of deflate stream");
        refill();
        for (int i = 0; i < nclen; ++i) {
            if (bitcnt < 3) {
                if (input_overrun()) return fail("unexpected end of deflate stream");
                refill();
            }
            clens[kClenOrder[i]] = (uint8_t)take(3);
        }
        if (input_overrun()) return fail("unexpected end of deflate stream");
        uint32_t ctab[128 + 19 * 2];
        if (!build_table(clens, 19, 7, ctab, (int)(sizeof(ctab) / sizeof(ctab[0])), [](int s) { return (uint32_t)s << kValShift; }))
            return fail("invalid code lengths set");
        int i = 0;
        while (i < nlit + ndist) {
            refill();
            if (input_overrun()) return fail("unexpected end of deflate stream");
            const uint32_t e = ctab[peek(7)];
            if (e & kKindInvalid) return fail("invalid code length code");
            drop((int)(e & 0xFF));
            const int sym = (int)(e >> kValShift);
            if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
            int rep;
            uint8_t val = 0;
            if (sym == 16) {
                if (i == 0) return fail("invalid bit length repeat");
                val = lens[i - 1];
                rep = 3 + (int)take(2);
            } else if (sym == 17) rep = 3 + (int)take(3);
            else rep = 11 + (int)take(7);
            if (i + rep > nlit + ndist) return fail("invalid bit length repeat");
            while (rep--) lens[i++] = val;
        }
        if (lens[256] == 0) return fail("invalid code -- missing end-of-block");
        // distance lengths follow the literal/length lengths: move them to a fixed place
        uint8_t dl[32] = {0};
        memcpy(dl, lens + nlit, (size_t)ndist);
        memset(lens + nlit, 0, (size_t)(288 - nlit));
        memcpy(lens + 288, dl, 32);
        nlit = 288; ndist = 32;
    } else {
        return fail("invalid block type");
    }
    if (!build_table(lens, nlit, kLitBits, lit, (int)(sizeof(lit) / sizeof(lit[0])), [](int s) -> uint32_t {
            if (s < 256) return kKindLiteral | ((uint32_t)s << kValShift);
            if (s == 256) return kKindEnd;
            if (s > 285) return kKindInvalid;
            return kKindBase | ((uint32_t)kLenBase[s - 257] << kValShift) | ((uint32_t)kLenExtra[s - 257] << kExtraShift);
        }))
        return fail("invalid literal/lengths set");
    if (!build_table(lens + 288, ndist, kDistBits, dist, (int)(sizeof(dist) / sizeof(dist[0])), [](int s) -> uint32_t {
            if (s > 29) return kKindInvalid;
            return kKindBase | ((uint32_t)kDistBase[s] << kValShift) | ((uint32_t)kDistExtra[s] << kExtraShift);
        }))
        return fail("invalid distances set");
    state = kHuffman;
    return true;
}

bool GzInflater::Impl::read_trailer()
{
    byte_align();
    in -= bitcnt >> 3;
    bitbuf = 0;
    bitcnt = 0;
    if (in_end - in < 8) return fail("unexpected end of file");
    const uint32_t want_crc = in[0] | (in[1] << 8) | (in[2] << 16) | ((uint32_t)in[3] << 24);
    const uint32_t want_len = in[4] | (in[5] << 8) | (in[6] << 16) | ((uint32_t)in[7] << 24);
    in += 8;
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
    s.in = data;
    s.in_end = data + n;
    s.bitbuf = 0;
    s.bitcnt = 0;
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
            if (s.input_overrun()) { s.fail("unexpected end of deflate stream"); return (size_t)-1; }
            if (!s.read_block_header()) return (size_t)-1;
            break;
        case Impl::kStored: {
            // byte aligned: hand back buffered whole bytes, then copy
            s.in -= s.bitcnt >> 3;
            s.bitbuf = 0;
            s.bitcnt = 0;
            size_t n = s.stored_left;
            if (s.in > s.in_end || (size_t)(s.in_end - s.in) < n) { s.fail("unexpected end of stored block"); return (size_t)-1; }
            if (o + n > o_limit) n = o < o_limit ? (size_t)(o_limit - o) : 0;
            memcpy(o, s.in, n);
            o += n;
            s.in += n;
            s.stored_left -= (uint32_t)n;
            if (s.stored_left == 0) s.state = s.last_block ? Impl::kTrailer : Impl::kBlockHeader;
            if (o >= o_limit && s.stored_left) { account(); return (size_t)(o - out); }
            break;
        }
        case Impl::kHuffman: {
            // the decoder state lives in locals inside the loop: byte stores through `o` may alias
            // anything reachable through `s`, which would force a reload after every literal
            const uint32_t *const lit = s.lit, *const dist = s.dist;
            const uint8_t *in = s.in;
            const uint8_t *const in_end = s.in_end;
            uint64_t bitbuf = s.bitbuf;
            int bitcnt = s.bitcnt;
            const char *err = nullptr;
            bool end_of_block = false;
#define MHX_REFILL()                                                                         \
    do {                                                                                     \
        uint64_t v_;                                                                         \
        memcpy(&v_, in, 8);                                                                  \
        bitbuf |= v_ << bitcnt;                                                              \
        in += (63 - bitcnt) >> 3;                                                            \
        bitcnt |= 56;                                                                        \
    } while (0)
#define MHX_DROP(n) do { const int n_ = (int)(n); bitbuf >>= n_; bitcnt -= n_; } while (0)
            while (o < o_limit) {
                MHX_REFILL();
                if (in - (bitcnt >> 3) > in_end) { err = "unexpected end of deflate stream"; break; }
                uint32_t e = lit[bitbuf & ((1u << kLitBits) - 1)];
                if (e & kKindSub) {
                    MHX_DROP(kLitBits);
                    e = lit[(e >> kValShift) + (uint32_t)(bitbuf & ((1ull << (e & 0xFF)) - 1))];
                }
                MHX_DROP(e & 0xFF);
                if (e & kKindLiteral) {
                    *o++ = (uint8_t)(e >> kValShift);
                    // a second and third literal usually fit the bits already buffered
                    e = lit[bitbuf & ((1u << kLitBits) - 1)];
                    if ((e & (kKindLiteral | kKindSub)) == kKindLiteral) {
                        MHX_DROP(e & 0xFF);
                        *o++ = (uint8_t)(e >> kValShift);
                        e = lit[bitbuf & ((1u << kLitBits) - 1)];
                        if ((e & (kKindLiteral | kKindSub)) == kKindLiteral) {
                            MHX_DROP(e & 0xFF);
                            *o++ = (uint8_t)(e >> kValShift);
                        }
                    }
                    continue;
                }
                if (e & kKindBase) {
                    const int le = (int)((e >> kExtraShift) & 15u);
                    const uint32_t len = (e >> kValShift) + (uint32_t)(bitbuf & ((1ull << le) - 1));
                    MHX_DROP(le);
                    if (bitcnt < 32) MHX_REFILL();
                    uint32_t d = dist[bitbuf & ((1u << kDistBits) - 1)];
                    if (d & kKindSub) {
                        MHX_DROP(kDistBits);
                        d = dist[(d >> kValShift) + (uint32_t)(bitbuf & ((1ull << (d & 0xFF)) - 1))];
                    }
                    if (!(d & kKindBase)) { err = "invalid distance code"; break; }
                    MHX_DROP(d & 0xFF);
                    const int de = (int)((d >> kExtraShift) & 15u);
                    const uint32_t distance = (d >> kValShift) + (uint32_t)(bitbuf & ((1ull << de) - 1));
                    MHX_DROP(de);
                    if ((size_t)(o - window_start) < distance) { err = "invalid distance too far back"; break; }
                    const uint8_t *src = o - distance;
                    uint8_t *const end = o + len;
                    if (distance >= 8) {
                        do { memcpy(o, src, 8); o += 8; src += 8; } while (o < end);
                    } else if (distance == 1) {
                        memset(o, *src, len);
                    } else {
                        while (o < end) *o++ = *src++;
                    }
                    o = end;
                    continue;
                }
                if (e & kKindEnd) { end_of_block = true; break; }
                err = "invalid literal/length code";
                break;
            }
#undef MHX_REFILL
#undef MHX_DROP
            s.in = in;
            s.bitbuf = bitbuf;
            s.bitcnt = bitcnt;
            if (err) { s.fail(err); return (size_t)-1; }
            if (end_of_block) s.state = s.last_block ? Impl::kTrailer : Impl::kBlockHeader;
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
