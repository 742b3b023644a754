// mhx_inflate_impl.h -- pieces of the DEFLATE decoder shared by the sequential inflater (mhx_inflate.cpp) and the
// parallel one (mhx_pinflate.cpp): bit reader, Huffman table construction, block header parsing and the symbol loop.
// Internal to libmhx.
#pragma once
#include <stdint.h>
#include <string.h>

namespace mhx {
namespace deflate {

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


// Table entries: bits 0..7 = bits to consume (or index bits of the sub-table), bit 8 literal, bit 9
// end of block, bit 10 sub-table link, bit 11 invalid, bit 12 length / distance base; bits 13..16 = number
// of extra bits that follow the code; bits 17..31 = the literal, the base length, the base distance or the
// sub-table offset.
struct Tables {
    uint32_t lit[(1 << kLitBits) + 288 * 16];
    uint32_t dist[(1 << kDistBits) + 32 * 128];
};

// Over-read discipline: the true read position is P = in - (bitcnt >> 3); a refill loads 8 bytes at
// `in` <= P + 7, i.e. touches bytes up to P + 14.  Every refill is preceded (at a distance of at most 2
// consumed bytes) by an overrun() test that pins P <= in_end, so no load reaches past in_end + 17 -- inside
// the GzInflater::kInputPad (64) zero bytes the caller guarantees behind the input.
struct BitReader {
    const uint8_t *in = nullptr, *in_end = nullptr; // in_end excludes the readable pad bytes
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    void refill()
    { // branch-free: valid while 8 bytes at `in` are readable (the buffer is padded)
        uint64_t v;
        memcpy(&v, in, 8);
        bitbuf |= v << bitcnt;
        in += (63 - bitcnt) >> 3;
        bitcnt |= 56;
    }
    uint32_t peek(int n) const { return (uint32_t)(bitbuf & ((1ull << n) - 1)); }
    void drop(int n) { bitbuf >>= n; bitcnt -= n; }
    uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
    bool overrun() const { return (in - (bitcnt >> 3)) > in_end; }
    void byte_align() { drop(bitcnt & 7); }
    // give back whole unread bytes: `in` is then the true position (only meaningful on a byte boundary)
    void unread() { in -= bitcnt >> 3; bitbuf = 0; bitcnt = 0; }
    // absolute bit position of the next unread bit, relative to `base`
    uint64_t bitpos(const uint8_t *base) const { return (uint64_t)(in - base) * 8 - (uint64_t)bitcnt; }
    void seek(const uint8_t *base, uint64_t bit)
    {
        in = base + (bit >> 3);
        bitbuf = 0;
        bitcnt = 0;
        refill();
        drop((int)(bit & 7));
    }
};

// Parses one block header at the reader's position.  type 0: stored block, *stored_len bytes follow (the reader is left
// byte-aligned, bits buffered); types 1/2: the tables are built.  Returns nullptr or the error text (zlib's wording).
inline const char *read_block_header(BitReader &r, Tables &t, bool *last_block, uint32_t *stored_len, uint32_t *type_out)
{
    r.refill();
    *last_block = r.take(1) != 0;
    const uint32_t type = r.take(2);
    *type_out = type;
    if (type == 0) {
        r.byte_align();
        if (r.overrun()) return "unexpected end of deflate stream";
        r.refill();
        const uint32_t len = r.take(16), nlen = r.take(16);
        // LEN/NLEN must lie inside the input: read from the zero pad they would pass the check as 0xFFFF/0x0000
        if (r.overrun()) return "unexpected end of deflate stream";
        if ((len ^ nlen) != 0xFFFFu) return "stored block length check failed";
        *stored_len = len;
        return nullptr;
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
        nlit = (int)r.take(5) + 257;
        ndist = (int)r.take(5) + 1;
        const int nclen = (int)r.take(4) + 4;
        if (nlit > 286 || ndist > 30) return "too many length or distance symbols";
        uint8_t clens[19] = {0};
        if (r.overrun()) return "unexpected end of deflate stream";
        r.refill();
        for (int i = 0; i < nclen; ++i) {
            if (r.bitcnt < 3) {
                if (r.overrun()) return "unexpected end of deflate stream";
                r.refill();
            }
            clens[kClenOrder[i]] = (uint8_t)r.take(3);
        }
        if (r.overrun()) return "unexpected end of deflate stream";
        uint32_t ctab[128 + 19 * 2];
        if (!build_table(clens, 19, 7, ctab, (int)(sizeof(ctab) / sizeof(ctab[0])), [](int s) { return (uint32_t)s << kValShift; }))
            return "invalid code lengths set";
        int i = 0;
        while (i < nlit + ndist) {
            r.refill();
            if (r.overrun()) return "unexpected end of deflate stream";
            const uint32_t e = ctab[r.peek(7)];
            if (e & kKindInvalid) return "invalid code length code";
            r.drop((int)(e & 0xFF));
            const int sym = (int)(e >> kValShift);
            if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
            int rep;
            uint8_t val = 0;
            if (sym == 16) {
                if (i == 0) return "invalid bit length repeat";
                val = lens[i - 1];
                rep = 3 + (int)r.take(2);
            } else if (sym == 17) rep = 3 + (int)r.take(3);
            else rep = 11 + (int)r.take(7);
            if (i + rep > nlit + ndist) return "invalid bit length repeat";
            while (rep--) lens[i++] = val;
        }
        if (lens[256] == 0) return "invalid code -- missing end-of-block";
        // distance lengths follow the literal/length lengths: move them to a fixed place
        uint8_t dl[32] = {0};
        memcpy(dl, lens + nlit, (size_t)ndist);
        memset(lens + nlit, 0, (size_t)(288 - nlit));
        memcpy(lens + 288, dl, 32);
        nlit = 288; ndist = 32;
    } else {
        return "invalid block type";
    }
    if (!build_table(lens, nlit, kLitBits, t.lit, (int)(sizeof(t.lit) / sizeof(t.lit[0])), [](int s) -> uint32_t {
            if (s < 256) return kKindLiteral | ((uint32_t)s << kValShift);
            if (s == 256) return kKindEnd;
            if (s > 285) return kKindInvalid;
            return kKindBase | ((uint32_t)kLenBase[s - 257] << kValShift) | ((uint32_t)kLenExtra[s - 257] << kExtraShift);
        }))
        return "invalid literal/lengths set";
    if (!build_table(lens + 288, ndist, kDistBits, t.dist, (int)(sizeof(t.dist) / sizeof(t.dist[0])), [](int s) -> uint32_t {
            if (s > 29) return kKindInvalid;
            return kKindBase | ((uint32_t)kDistBase[s] << kValShift) | ((uint32_t)kDistExtra[s] << kExtraShift);
        }))
        return "invalid distances set";
    return nullptr;
}

// The symbol loop of a Huffman block, on output elements of type T: uint8_t for the plain decoder, uint16_t for the
// parallel decoder's symbolic pass (bytes, or markers >= 0x8000 that stand for bytes of the not yet known 32 KiB in front
// of a segment).  Decodes until `o_limit` is reached (may overshoot by < 320 elements), the block ends or an error occurs.
// [window_start, o) is the output so far (a match may reach back 32 KiB into it).
enum BlockStatus { kBlockEnd, kBlockLimit, kBlockError };
template <class T>
inline BlockStatus huffman_block(BitReader &r, const Tables &t, T *&o_ref, T *const o_limit, const T *window_start, const char **err_out)
{
    // the decoder state lives in locals inside the loop: stores through `o` may alias
    // anything reachable through the reader, which would force a reload after every literal
    const uint32_t *const lit = t.lit, *const dist = t.dist;
    const uint8_t *in = r.in;
    const uint8_t *const in_end = r.in_end;
    uint64_t bitbuf = r.bitbuf;
    int bitcnt = r.bitcnt;
    T *o = o_ref;
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
            *o++ = (T)(e >> kValShift);
            // a second and third literal usually fit the bits already buffered
            e = lit[bitbuf & ((1u << kLitBits) - 1)];
            if ((e & (kKindLiteral | kKindSub)) == kKindLiteral) {
                MHX_DROP(e & 0xFF);
                *o++ = (T)(e >> kValShift);
                e = lit[bitbuf & ((1u << kLitBits) - 1)];
                if ((e & (kKindLiteral | kKindSub)) == kKindLiteral) {
                    MHX_DROP(e & 0xFF);
                    *o++ = (T)(e >> kValShift);
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
            const T *src = o - distance;
            T *const end = o + len;
            if (distance * sizeof(T) >= 8) {
                do { memcpy(o, src, 8); o += 8 / sizeof(T); src += 8 / sizeof(T); } while (o < end);
            } else if (distance == 1) {
                const T v = *src;
                for (T *p = o; p < end; ++p) *p = v;
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
    r.in = in;
    r.bitbuf = bitbuf;
    r.bitcnt = bitcnt;
    o_ref = o;
    if (err) { *err_out = err; return kBlockError; }
    return end_of_block ? kBlockEnd : kBlockLimit;
}

} // namespace deflate
} // namespace mhx
