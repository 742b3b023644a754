// mhx_tile.h -- per-tile logic of the sketch kernel, written as host+device inline
// functions: mhx_kernels.hip strings the phases together with __syncthreads(); the CPU
// phase emulator (tests/emul/tile_emul.cpp) runs the very same functions thread by
// thread so that indexing and bit logic are checked without a GPU.
//
// What one tile does (replaces mash's kseq_read + addMinHashes + getHash hot loop,
// Mash 2.x Sketch.cpp; reached from /root/reference/auriclass/classes.py:576-596,696-713):
//   stage     16 KiB (+64 B halo) of the byte stream into LDS, 16 B per lane, coalesced
//   classify  per byte: newline?                                 -> bit mask (SIMD-in-register)
//   phase     FASTQ: newline prefix -> line number mod 4 == 1 marks sequence lines
//   runs      candidate k-mer starts = K bytes inside one line   -> 1 bit per position
//   compact   groups of 8 start positions with any candidate     -> LDS work list
//   work      each lane takes a group: 8 windows share one 28..39 byte register chunk;
//             canonical strand by big-endian compare, MurmurHash3_x64_128(seed 42) up to its
//             last two steps, a 32-bit necessary test against the global threshold; the few
//             windows that pass finish the hash, are checked base by base (A/C/G/T, either
//             case -- mash skips every window holding anything else) and go into the table.
// The base check is deferred on purpose: one window in 10^4..10^6 is ever a candidate, so
// testing every byte of the stream for A/C/G/T up front cost more than hashing the rare
// window that holds an N (its hash is garbage, it passes the threshold as rarely as any
// other, and the exact check then drops it).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MHX_HD __host__ __device__ __forceinline__
#else
#define MHX_HD inline
struct uint4 { uint32_t x, y, z, w; };
#endif

#include "mhx_device_consts.h"

// rare branches are laid out of line: the hot path falls through instead of jumping over the cold block
#define MHX_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace mhx {

constexpr int kMapWords = (kTileBytes + kHaloBytes) / 32 + 2; // words of a 1-bit-per-byte map of tile + halo (+2 look-ahead)
// The queue form of the kernel (large sketches, early launches; process_group_regs<K, true>): windows whose partial
// hash passes the admission test are not finished where they are found -- one candidate lane would take the other 63
// through the cold code (hash tail, exact compare, base check, two atomics: ~100 instructions), and with s = 50 000 (one
// window in 230 below T) that happens in every fourth wave-iteration -- but queued as (group << 3) | window in the part of
// the work list the tile's items leave free (list[nitems ..)) and finished together after the hash loop, one per lane
// (process_deferred).  A window that finds the queue full is finished at once.
struct TileSmem {
    uint4 bytes[(kTileBytes + kHaloBytes) / 16];      // staged stream bytes
    uint32_t valid[kGroupsPerTile / 4];                // byte g = candidate-start mask of group g
    // One area, three tenants with disjoint lifetimes: the newline map (classify -> good-map phase) and the
    // good map (good-map phase -> candidate starts) side by side, then the work list (compaction -> hash loop).
    // 22.7 KB of LDS per workgroup in all: seven workgroups per CU.
    union {
        uint16_t list[kGroupsPerTile];                 // compacted work list (group ids)
        uint32_t maps[2 * kMapWords];
    };
    uint32_t cnt[16];                                  // wave partials of the two workgroup scans
    uint32_t misc[8];                                  // 0: line base, 1: #items, 2: tile id, 3: k-mers, 4: inserts, 5: long records, 7: #queued candidates
};

MHX_HD uint32_t *tile_nlmap(TileSmem &sm) { return sm.maps; }             // 1 bit per byte: newline
MHX_HD uint32_t *tile_good(TileSmem &sm) { return sm.maps + kMapWords; } // 1 bit per byte: may be covered by a k-mer

// per-thread state carried between phases (registers on the GPU)
struct ThreadState {
    uint32_t nl[kWordsPerThread];  // newline mask of this thread's 32-byte words (bytes outside the span cleared)
    uint32_t in[kWordsPerThread];  // bytes of those words that lie inside the span
    uint32_t hnl[2], hin[2];       // thread 0 only: the two halo words
    uint32_t nlcount;
};

// ---- byte-parallel helpers ----------------------------------------------------------
MHX_HD uint32_t zero_byte_flags(uint32_t y)
{ // 0x80 in every byte of y that is zero (exact, no cross-byte carries)
    uint32_t t = (y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | y | 0x7F7F7F7Fu);
}
MHX_HD uint32_t flags_to_nibble(uint32_t f)
{ // bits 7,15,23,31 -> bits 0..3
    return (f * 0x00204081u) >> 28;
}
MHX_HD uint32_t perm_lut(uint32_t lut, uint32_t sel)
{ // out.byte[i] = lut.byte[sel.byte[i]]  (selectors 0..3)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(lut, lut, sel);
#else
    uint32_t o = 0;
    for (int i = 0; i < 4; ++i) o |= ((lut >> (8 * ((sel >> (8 * i)) & 3))) & 0xFFu) << (8 * i);
    return o;
#endif
}
// 2-bit index of a (case-folded) byte: bits 1..2, which tell A, C, G, T apart (0, 1, 3, 2); any other
// byte lands on one of the four as well and is exposed by the comparison with the table entry
MHX_HD uint32_t base_index(uint32_t u) { return (u >> 1) & 0x03030303u; }
constexpr uint32_t kLutBase = 0x47544341u; // index -> 'A','C','T','G'
constexpr uint32_t kLutComp = 0x43414754u; // index -> complement: 'T','G','A','C'

// low 32 bits of {hi:lo} >> (8 * byte_shift), byte_shift in 0..3.  On the device this must be
// the v_alignbyte/v_alignbit instruction itself: written as a C shift-or, LLVM turns the
// pattern into an unaligned load from a stack copy of the register array (scratch traffic).
MHX_HD uint32_t funnel(uint32_t hi, uint32_t lo, int byte_shift)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)byte_shift);
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * byte_shift));
#endif
}
// value the optimiser must treat as unknown at this point (pins cold-path work behind its branch)
MHX_HD uint32_t opaque(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}
MHX_HD uint32_t funnel_bits(uint32_t hi, uint32_t lo, uint32_t bit_shift)
{ // bit_shift in 0..31
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, bit_shift);
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> bit_shift);
#endif
}

// 0x80 in every byte of v that equals the byte replicated in `pattern` (exact).  One instruction less than
// zero_byte_flags(v ^ pattern): the xor-ed word itself is never needed, only its low seven bits per byte (one v_bitop3:
// (v ^ pattern) & 0x7F7F7F7F) and its top bits, which for patterns below 0x80 are v's own top bits.
MHX_HD uint32_t byte_eq_flags(uint32_t v, uint32_t pattern)
{
    const uint32_t t = ((v ^ pattern) & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | v) & 0x80808080u;
}
// flags (0x80 per byte) of two consecutive dwords -> 8 mask bits, low dword first, with ONE multiplication: the first
// dword's flags move to bit 3 of each byte, the second's stay at bit 7; times 0x00204081 the eight flags land in bits
// 24..31 in byte order (the stray partial products fall on distinct bits below 24 or beyond 31, so nothing carries).
MHX_HD uint32_t flags_to_byte(uint32_t f_lo, uint32_t f_hi) { return (((f_lo >> 4) | f_hi) * 0x00204081u) >> 24; }

// newline mask of one 32-byte word (8 dwords at p)
MHX_HD uint32_t newline_mask(const uint32_t *p)
{
    uint32_t n = 0;
#pragma unroll
    for (int d = 0; d < 8; d += 2)
        n |= flags_to_byte(byte_eq_flags(p[d], 0x0A0A0A0Au), byte_eq_flags(p[d + 1], 0x0A0A0A0Au)) << (4 * d);
    return n;
}
// A/C/G/T (either case) mask of the four bytes of a dword, as flags 0x80 per byte
MHX_HD uint32_t acgt_flags(uint32_t v)
{
    const uint32_t u = v & 0xDFDFDFDFu;
    return zero_byte_flags(perm_lut(kLutBase, base_index(u)) ^ u);
}

// bits of a 32-byte word starting at absolute offset A that lie inside [begin, end)
MHX_HD uint32_t inrange_mask(uint64_t A, uint64_t begin, uint64_t end)
{
    uint32_t m = 0xFFFFFFFFu;
    if (begin > A) { uint64_t lo = begin - A; m = lo >= 32 ? 0u : (m << lo); }
    if (end < A + 32) { uint64_t hi = end > A ? end - A : 0; m &= hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u); }
    return m;
}

// Counts the records mash counts: those whose sequence line holds at least k bytes (feeds the
// "[N seqs]" comment of the .msh).  A sequence line starts right after the newline that ends a
// header; it is long enough iff none of its first k bytes is a newline and they all lie in the span.
struct LineLenCheck {
    const uint32_t *nlmap; // one newline bit per tile byte (+ the 64 halo bytes), span-masked
    uint64_t tile_off;     // absolute offset of tile byte 0
    uint64_t end;          // span end
    uint32_t k;
    uint32_t count;
};

// Bytes of lines whose index is 1 (mod 4), newline bytes excluded.  `line` is the line
// index at the first byte of the word.  Also verifies the 4-line layout: the line after a
// newline must start with '@' (index 0 mod 4) or '+' (index 2 mod 4).  tile_bytes/word_off
// locate the word inside the staged tile for that look-ahead (nullptr: skip the check).
MHX_HD uint32_t seqline_mask(uint32_t nl, uint32_t line, const uint8_t *tile_bytes, uint32_t word_off,
                             uint32_t check_limit, bool &bad_format, LineLenCheck *lc = nullptr)
{
    uint32_t m = 0, cur = 0xFFFFFFFFu, pend = 0;
    while (nl) {
        const uint32_t b = nl & (0u - nl);
        const uint32_t below = b - 1u;
        if ((line & 3u) == 1u) m |= cur & below;
        cur = ~(below | b);
        ++line;
        if (tile_bytes) {
            const uint32_t pos = word_off + (uint32_t)__builtin_ctz(b) + 1u;
            if (pos < check_limit) {
                const uint8_t c = tile_bytes[pos];
                if (((line & 3u) == 0u && c != '@') || ((line & 3u) == 2u && c != '+')) bad_format = true;
            }
            if (lc && (line & 3u) == 1u) pend |= b; // a sequence line starts behind this newline: measured below
        }
        nl &= nl - 1u;
    }
    // the length test of the sequence lines that start in this word, outside the per-newline loop (which
    // every lane of the wave sits through once per newline of the busiest lane)
    while (pend) {
        const uint32_t pos = word_off + (uint32_t)__builtin_ctz(pend) + 1u;
        const uint32_t wq = pos >> 5, sh = pos & 31u;
        const uint32_t window = funnel_bits(lc->nlmap[wq + 1], lc->nlmap[wq], sh); // newline bits of bytes pos..pos+31
        const uint32_t first_k = lc->k >= 32 ? 0xFFFFFFFFu : ((1u << lc->k) - 1u);
        if ((window & first_k) == 0 && lc->tile_off + pos + lc->k <= lc->end) {
            // a '\r' that ends the line is not part of the sequence (kseq drops it): if the k-th byte
            // is a CR followed by the newline (or by the end of the stream) the line has k-1 bases
            const uint32_t after = pos + lc->k;
            const bool ends_here = lc->tile_off + after >= lc->end || ((lc->nlmap[after >> 5] >> (after & 31u)) & 1u);
            if (!(tile_bytes[after - 1u] == '\r' && ends_here)) ++lc->count;
        }
        pend &= pend - 1u;
    }
    if ((line & 3u) == 1u) m |= cur;
    return m;
}

// bit p of the result is set iff bits p..p+K-1 of the window w[0..NWD] are all set (the
// last word is look-ahead only; NWD result words are returned).
// R_{2n} = R_n & (R_n >> n); R_{n+1} = R_n & (w >> n).
template <int K, int NWD> MHX_HD void run_starts(const uint32_t (&w)[NWD + 1], uint32_t (&out)[NWD])
{
    constexpr int N1 = NWD + 1;
    uint32_t cur[N1];
#pragma unroll
    for (int i = 0; i < N1; ++i) cur[i] = w[i];
    int len = 1;
    constexpr int top = K >= 32 ? 5 : K >= 16 ? 4 : K >= 8 ? 3 : K >= 4 ? 2 : K >= 2 ? 1 : 0;
#pragma unroll
    for (int b = top - 1; b >= 0; --b) {
        // double
        {
            uint32_t sh[N1];
#pragma unroll
            for (int i = 0; i < N1; ++i) {
                const uint32_t hi = (i + 1 < N1) ? cur[i + 1] : 0u;
                sh[i] = (uint32_t)(((((uint64_t)hi) << 32) | cur[i]) >> len);
            }
#pragma unroll
            for (int i = 0; i < N1; ++i) cur[i] &= sh[i];
            len *= 2;
        }
        if ((K >> b) & 1) {
            uint32_t sh[N1];
#pragma unroll
            for (int i = 0; i < N1; ++i) {
                const uint32_t hi = (i + 1 < N1) ? w[i + 1] : 0u;
                sh[i] = (uint32_t)(((((uint64_t)hi) << 32) | w[i]) >> len);
            }
#pragma unroll
            for (int i = 0; i < N1; ++i) cur[i] &= sh[i];
            len += 1;
        }
    }
#pragma unroll
    for (int i = 0; i < NWD; ++i) out[i] = cur[i];
}

// ---- MurmurHash3_x64_128, seed 42, first 8 output bytes (mash getHash) ----------------
// 64-bit values are handled so that hipcc keeps each step in its cheapest form on gfx950:
//  * a product is made opaque before it is rotated: left to itself the compiler turns rotl(k * c, r) into a SECOND
//    64x64 multiplication by (c << r) plus the shifted high word (10 instructions where 4 + 2 do);
//  * rotations and the fmix shift-xors are written on the 32-bit halves (2 v_alignbit / v_lshrrev + v_xor);
//  * h * 5 + c is one v_mad_u64_u32 for the low word and a shift-add for the high word.
MHX_HD uint64_t opaque64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(v));
#endif
    return v;
}
MHX_HD uint64_t make64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
template <int R> MHX_HD uint64_t rotl64(uint64_t x)
{
    static_assert(R > 0 && R < 64 && R != 32, "rotation amount");
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (R < 32) return make64(funnel_bits(lo, hi, 32 - R), funnel_bits(hi, lo, 32 - R));
    return make64(funnel_bits(hi, lo, 64 - R), funnel_bits(lo, hi, 64 - R));
}
MHX_HD uint64_t xorshift33(uint64_t k)
{ // k ^= k >> 33: only the low word changes
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    return make64(lo ^ (hi >> 1), hi);
}
MHX_HD uint64_t times5_plus(uint64_t h, uint32_t c)
{
    const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
    const uint64_t t = (uint64_t)lo * 5u + c;
    return make64((uint32_t)t, ((hi << 2) + hi) + (uint32_t)(t >> 32));
}
// x * C mod 2^64 for a constant C.  hipcc's own sequence is v_mad_u64_u32 (lo * C_lo, 64 bits) + 2 x v_mul_lo_u32 (the cross
// terms) + v_add3_u32: four quarter-rate VALU instructions (4 cycles each per wave64 in isolation,
// profiles/r02_valu_class_rates_microbench.txt).  -DMHX_ASM_MUL64 selects an experiment of round 3 instead: three of them
// plus one full-rate add -- the cross sum lands in the ODD half of a register pair whose even half holds zero, and the
// last v_mad_u64_u32 takes that pair, (cross << 32), as its 64-bit addend.  The compiler cannot be talked into this by
// itself (it builds {0, cross} with two v_mov per multiply), so the four instructions are one asm block on the fixed pair
// v[70:71]; `zero` is the variable that lives in v70, threaded through every call of one hash.  A block of four with no
// hazard inside gets no s_nop brackets (round 2 wrapped single instructions and drowned in them).  Measured in the
// kernel (profiles/r03_hash_loop_asm_multiply_ab.txt): k=21 -0.3 %, k=27 +2.0 % -- no gain: 80 v_add3 became 80
// v_add_u32 per group of 8 windows, but 13 v_mov and 14 s_nop came with the fixed registers, and inside the kernel an
// instruction costs its ~3.5 issue cycles whatever its class.  The count is what matters, and four
// instructions per multiply is the floor: the 64-bit addend of v_mad_u64_u32 must be an even-aligned pair, and the value
// that has to go into its ODD half is produced in an even one.  The compiler's sequence stays the default.
template <uint64_t C> MHX_HD uint64_t mul64c(uint64_t x, uint32_t &zero)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(MHX_ASM_MUL64)
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    uint64_t p, carry;
    uint32_t y;
    asm("v_mul_lo_u32 v71, %[lo], %[chi]\n\t"
        "v_mul_lo_u32 %[y], %[hi], %[clo]\n\t"
        "v_add_u32_e32 v71, v71, %[y]\n\t"
        "v_mad_u64_u32 %[p], %[c], %[lo], %[clo], v[70:71]"
        : [p] "=&v"(p), [c] "=&s"(carry), [y] "=&v"(y), "+{v70}"(zero)
        : [lo] "v"(lo), [hi] "v"(hi), [clo] "s"((uint32_t)C), [chi] "s"((uint32_t)(C >> 32))
        : "v71");
    return p;
#else
    (void)zero;
    return opaque64(x * C);
#endif
}
// fmix64 without its last step (k ^= k >> 33 changes only the low word, see Murmur3Tail::finish)
MHX_HD uint64_t fmix64_head(uint64_t k, uint32_t &zero)
{
    k = xorshift33(k);
    k = mul64c<0xff51afd7ed558ccdull>(k, zero);
    k = xorshift33(k);
    return mul64c<0xc4ceb9fe1a85ec53ull>(k, zero);
}
// The hash up to its last two steps: h = xorshift33(a) + xorshift33(b).  The high word of h is
// hi(a) + hi(b) or that plus one, so `h <= T` implies hi(a) + hi(b) + 1 <= hi(T) + 1 (mod 2^32, see
// admission_limit): one 32-bit add and one compare reject all but ~T / 2^64 of the windows without the
// two shift-xors and the 64-bit add.
struct Murmur3Tail {
    uint64_t a, b;
    MHX_HD uint64_t finish() const { return xorshift33(a) + xorshift33(b); }
    MHX_HD uint32_t low32() const { return (uint32_t)xorshift33(a) + (uint32_t)xorshift33(b); } // 32-bit hashes (k <= 16)
    MHX_HD uint32_t high_bound() const { return (uint32_t)(a >> 32) + (uint32_t)(b >> 32) + 1u; }
};
// largest value of Murmur3Tail::high_bound() a hash <= T can have
MHX_HD uint32_t admission_limit(uint64_t T)
{
    const uint32_t th = (uint32_t)(T >> 32);
    return th == 0xFFFFFFFFu ? th : th + 1u;
}
// w: the K bytes as little-endian dwords, bytes beyond K zero
template <int K> MHX_HD Murmur3Tail murmur3_core(const uint32_t (&w)[8])
{
    constexpr uint64_t c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    constexpr int NBLK = K / 16, TAIL = K & 15;
    uint64_t h1 = 42, h2 = 42;
    uint32_t zero = 0; // see mul64c
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        uint64_t k1 = make64(w[4 * b], w[4 * b + 1]);
        uint64_t k2 = make64(w[4 * b + 2], w[4 * b + 3]);
        k1 = rotl64<31>(mul64c<c1>(k1, zero)); k1 = mul64c<c2>(k1, zero); h1 ^= k1;
        h1 = rotl64<27>(h1); h1 += h2; h1 = times5_plus(h1, 0x52dce729u);
        k2 = rotl64<33>(mul64c<c2>(k2, zero)); k2 = mul64c<c1>(k2, zero); h2 ^= k2;
        h2 = rotl64<31>(h2); h2 += h1; h2 = times5_plus(h2, 0x38495ab5u);
    }
    if (TAIL > 8) {
        uint64_t k2 = make64(w[(4 * NBLK + 2) & 7], w[(4 * NBLK + 3) & 7]);
        k2 = rotl64<33>(mul64c<c2>(k2, zero)); k2 = mul64c<c1>(k2, zero); h2 ^= k2;
    }
    if (TAIL > 0) {
        uint64_t k1 = make64(w[(4 * NBLK) & 7], w[(4 * NBLK + 1) & 7]);
        k1 = rotl64<31>(mul64c<c1>(k1, zero)); k1 = mul64c<c2>(k1, zero); h1 ^= k1;
    }
    h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
    h1 += h2; h2 += h1;
    const uint64_t a = fmix64_head(h1, zero), b = fmix64_head(h2, zero);
    return Murmur3Tail{a, b};
}
template <int K> MHX_HD uint64_t murmur3_h1(const uint32_t (&w)[8]) { return murmur3_core<K>(w).finish(); }

// ---- phases -------------------------------------------------------------------------

// P1: stage the tile.  chunk c (16 B) of the tile is loaded by thread c % 256.
MHX_HD void phase_stage(TileSmem &sm, int tid, const uint8_t *base, uint64_t tile_off, uint64_t end)
{
    const uint64_t lim = (end + 15) & ~(uint64_t)15; // last readable 16-byte chunk boundary
    constexpr int kChunks = kTileBytes / 16;          // 1024
#ifndef MHX_STAGE_SERIAL
    // tile and halo inside the readable span (wave-uniform; every tile but the last): all loads of a lane are issued
    // before the first LDS write -- behind the per-chunk bound test below hipcc waits for each load (s_waitcnt vmcnt(0))
    // before it issues the next, four HBM round trips in a row at the head of every tile
    if (tile_off + (uint64_t)(kTileBytes + kHaloBytes) <= lim) {
        const uint4 *src = reinterpret_cast<const uint4 *>(base + tile_off) + tid;
        uint4 v[kChunks / kBlock], h = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < kChunks / kBlock; ++j) v[j] = src[j * kBlock];
        if (tid < kHaloBytes / 16) h = src[kChunks];
#pragma unroll
        for (int j = 0; j < kChunks / kBlock; ++j) sm.bytes[j * kBlock + tid] = v[j];
        if (tid < kHaloBytes / 16) sm.bytes[kChunks + tid] = h;
        return;
    }
#endif
#pragma unroll
    for (int j = 0; j < kChunks / kBlock; ++j) {
        const int c = j * kBlock + tid;
        const uint64_t off = tile_off + (uint64_t)c * 16;
        uint4 v = {0, 0, 0, 0};
        if (off < lim) v = *reinterpret_cast<const uint4 *>(base + off);
        sm.bytes[c] = v;
    }
    if (tid < kHaloBytes / 16) {
        const int c = kChunks + tid;
        const uint64_t off = tile_off + (uint64_t)c * 16;
        uint4 v = {0, 0, 0, 0};
        if (off < lim) v = *reinterpret_cast<const uint4 *>(base + off);
        sm.bytes[c] = v;
    }
}

// P2a: newline mask of this thread's 128 bytes (and, for thread 0, of the halo).  `interior`: the tile and its
// halo lie inside the span, so no byte has to be masked out (wave-uniform: all tiles but the first and the last)
MHX_HD void phase_classify(TileSmem &sm, int tid, ThreadState &st, uint64_t tile_off, uint64_t begin, uint64_t end, bool interior)
{
    const uint32_t *p = reinterpret_cast<const uint32_t *>(sm.bytes) + tid * (kBytesPerThread / 4);
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kWordsPerThread; ++w) {
        st.in[w] = interior ? 0xFFFFFFFFu : inrange_mask(tile_off + (uint64_t)tid * kBytesPerThread + 32u * w, begin, end);
        st.nl[w] = newline_mask(p + 8 * w) & st.in[w];
        total += (uint32_t)__builtin_popcount(st.nl[w]);
        tile_nlmap(sm)[tid * kWordsPerThread + w] = st.nl[w];
    }
    st.nlcount = total;
    if (tid == 0) {
        const uint32_t *h = reinterpret_cast<const uint32_t *>(sm.bytes) + kTileBytes / 4;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            st.hin[w] = interior ? 0xFFFFFFFFu : inrange_mask(tile_off + kTileBytes + 32u * w, begin, end);
            st.hnl[w] = newline_mask(h + 8 * w) & st.hin[w];
            tile_nlmap(sm)[kTileBytes / 32 + w] = st.hnl[w];
        }
    }
}


// FASTQ line phase of a tile WITHOUT its predecessors: the line index (mod 4) of the tile's first byte, or 4 if the tile
// cannot tell.  A header is the only line that starts with '@', is followed by a line that does not, and then by a line
// that starts with '+' (a quality line may start with '@', but the line after it is a header and starts with '@' too), so
// the first such triple among the tile's first six line starts fixes the phase: with i newlines in front of the header
// line the tile's first byte lies in line -i (mod 4).  Needs six newlines within the staged, in-span bytes -- reads of up
// to ~2.7 kb; tiles of longer lines take the look-back pass instead.  The per-record layout check of phase_good still
// runs over every line, so a phase derived from a malformed file is caught there.
MHX_HD uint32_t phase_selfsync(TileSmem &sm, uint32_t check_limit)
{
    const uint32_t *nlmap = tile_nlmap(sm);
    const uint8_t *b = reinterpret_cast<const uint8_t *>(sm.bytes);
    uint32_t pos[6];
    int n = 0;
    for (int w = 0; w < kTileBytes / 32 + 2 && n < 6; ++w) {
        uint32_t m = nlmap[w];
        while (m && n < 6) {
            pos[n++] = 32u * (uint32_t)w + (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
        }
    }
    for (int i = 0; i + 2 < n && i < 4; ++i) {
        const uint32_t p0 = pos[i] + 1u, p1 = pos[i + 1] + 1u, p2 = pos[i + 2] + 1u;
        if (p2 >= check_limit) break;
        if (b[p0] == '@' && b[p1] != '@' && b[p1] != '+' && b[p2] == '+') return (4u - ((uint32_t)(i + 1) & 3u)) & 3u;
    }
    return 4u;
}

// What a tile leaves behind for the chain check: the line phase of its first byte and of the first byte of the next
// tile.  phase_verify_kernel compares neighbours, so that the phases the tiles found by themselves are the ones a
// running line count from the start of the span gives -- a file on which they are not (a record cut short in front of
// a tile border, which kseq reads as a record without qualities) is flagged and goes to the general parser.
constexpr uint8_t kPhaseKnown = 0x10;
MHX_HD uint8_t phase_record(uint32_t start_phase, uint32_t tile_lines)
{
    return (uint8_t)(kPhaseKnown | (start_phase & 3u) | (((start_phase + tile_lines) & 3u) << 2));
}
MHX_HD bool phase_chain_broken(uint8_t prev, uint8_t cur)
{
    return (prev & kPhaseKnown) && (cur & kPhaseKnown) && ((prev >> 2) & 3u) != (cur & 3u);
}

// P2c: bytes a k-mer may cover -> good map: FASTQ: the bytes of sequence lines; sequence stream: everything but the
// record separators.  (Whether they are A/C/G/T is checked for the few windows that pass the threshold.)
// returns the number of records of this thread's bytes whose sequence line holds >= k bytes
template <bool FASTQ>
MHX_HD uint32_t phase_good(TileSmem &sm, int tid, const ThreadState &st, uint32_t line_base, uint32_t excl,
                           uint32_t tile_total, uint32_t check_limit, bool &bad_format, uint64_t tile_off, uint64_t end, uint32_t k)
{
    const uint8_t *tb = reinterpret_cast<const uint8_t *>(sm.bytes);
    LineLenCheck lc{tile_nlmap(sm), tile_off, end, k, 0u};
    uint32_t line = line_base + excl;
#pragma unroll
    for (int w = 0; w < kWordsPerThread; ++w) {
        uint32_t g;
        if (FASTQ) {
            g = st.in[w] & seqline_mask(st.nl[w], line, tb, (uint32_t)tid * kBytesPerThread + 32u * w, check_limit, bad_format, &lc);
            line += (uint32_t)__builtin_popcount(st.nl[w]);
        } else {
            g = st.in[w] & ~st.nl[w];
        }
        tile_good(sm)[tid * kWordsPerThread + w] = g;
    }
    if (tid == 0) {
        uint32_t hl = line_base + tile_total;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            uint32_t g;
            if (FASTQ) {
                bool ignore = false;
                g = st.hin[w] & seqline_mask(st.hnl[w], hl, nullptr, 0, 0, ignore);
                hl += (uint32_t)__builtin_popcount(st.hnl[w]);
            } else {
                g = st.hin[w] & ~st.hnl[w];
            }
            tile_good(sm)[kTileBytes / 32 + w] = g;
        }
        tile_good(sm)[kTileBytes / 32 + 2] = 0;
        tile_good(sm)[kTileBytes / 32 + 3] = 0;
    }
    return lc.count;
}

// P3: valid k-mer starts of this thread's positions -> sm.valid, #items -> sm.cnt
template <int K> MHX_HD uint32_t phase_runs(TileSmem &sm, int tid, uint32_t &items_out)
{
    constexpr int NWD = kWordsPerThread;
    uint32_t w[NWD + 1], v[NWD];
#pragma unroll
    for (int i = 0; i < NWD + 1; ++i) w[i] = tile_good(sm)[tid * NWD + i];
    run_starts<K, NWD>(w, v);
    uint32_t items = 0, kmers = 0;
#pragma unroll
    for (int i = 0; i < NWD; ++i) {
        sm.valid[tid * NWD + i] = v[i];
        kmers += (uint32_t)__builtin_popcount(v[i]);
        items += 4u - (uint32_t)__builtin_popcount(zero_byte_flags(v[i]));
    }
    items_out = items;
    return kmers;
}

// P4: append this thread's non-empty groups to the work list
MHX_HD void phase_compact(TileSmem &sm, int tid, uint32_t excl)
{
    uint32_t pos = excl;
#pragma unroll
    for (int i = 0; i < kWordsPerThread; ++i) {
        const uint32_t v = sm.valid[tid * kWordsPerThread + i];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if ((v >> (8 * b)) & 0xFFu) sm.list[pos++] = (uint16_t)((tid * kWordsPerThread + i) * 4 + b);
    }
    if (tid == kBlock - 1) sm.misc[1] = pos;
}

// ASCII complement of four folded bases at once: bits 1..2 of a base tell 'A','C','T','G' apart (0,1,2,3) and the
// complement comes out of a 4-byte table in one v_perm_b32.  Bytes that are not A/C/G/T come out as one of
// the four letters as well; a window that holds one never reaches the table (process_deferred checks the bases).
MHX_HD uint32_t complement4(uint32_t u) { return perm_lut(kLutComp, base_index(u)); }

// 64 bits starting at byte offset `off` (compile-time) of the dword array a
template <int OFF, int N> MHX_HD uint64_t load64(const uint32_t (&a)[N])
{
    constexpr int q = OFF / 4, sh = OFF % 4;
    static_assert(q + (sh ? 2 : 1) < N, "load64 out of range");
    const uint32_t lo = sh ? funnel(a[q + 1], a[q], sh) : a[q];
    const uint32_t hi = sh ? funnel(a[q + 2], a[q + 1], sh) : a[q + 1];
    return ((uint64_t)hi << 32) | lo;
}

// memcmp(fwd, rc, K) > 0, on the already extracted little-endian words (slow, exact)
template <int NW> MHX_HD bool rc_is_smaller_full(const uint32_t (&wf)[8], const uint32_t (&wr)[8])
{
    bool rc_less = false, decided = false;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t a = __builtin_bswap32(wf[i]), b = __builtin_bswap32(wr[i]);
        rc_less = decided ? rc_less : (b < a);
        decided = decided || (a != b);
    }
    return rc_less;
}

// K bytes at byte offset OFF of the dword array src, as zero-padded little-endian words
template <int K, int OFF, int N> MHX_HD void extract_words(const uint32_t (&src)[N], uint32_t (&w)[8])
{
    constexpr int NW = (K + 3) / 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = i < NW ? funnel(src[OFF / 4 + i + 1], src[OFF / 4 + i], OFF % 4) : 0u;
    if (K % 4) w[NW - 1] &= (1u << (8 * (K % 4))) - 1u;
}

// One window: the K bytes of its canonical strand as little-endian words.  U = folded forward chunk, R = its
// reverse complement, Wr = the forward chunk byte-reversed, Cc = the forward chunk complemented (the last two
// only feed the strand decision).
template <int K, int J, int ND>
MHX_HD void canonical_words(const uint32_t (&U)[ND + 1], const uint32_t (&R)[ND + 1], const uint32_t (&Wr)[ND + 1],
                            const uint32_t (&Cc)[ND + 1], uint32_t (&w)[8])
{
    constexpr int NW = (K + 3) / 4;
    constexpr int OF = J;               // forward window starts at U byte OF
    constexpr int OR = ND * 4 - K - J;  // its reverse complement starts at R byte OR
    if constexpr (K >= 8) {
        // memcmp(fwd, rc) compares big-endian; the first 8 bases of either strand, most
        // significant first, are 8 little-endian bytes of Wr resp. Cc.  Equal first 8 bases
        // (4^-8 of the windows) fall back to the full comparison.
        const uint64_t top_f = load64<ND * 4 - 8 - J>(Wr);
        const uint64_t top_r = load64<J + K - 8>(Cc);
        bool rc = top_r < top_f;
        if (MHX_UNLIKELY(K > 8 && top_r == top_f)) { // cold: exact comparison
            // the copies go through an opaque barrier so that the compiler cannot hoist the
            // word extraction of this 4^-8 case in front of the branch (it did: ~19 wasted
            // instructions per window on the hot path)
            uint32_t Uc[ND + 1], Rc[ND + 1];
#pragma unroll
            for (int i = 0; i < ND + 1; ++i) { Uc[i] = opaque(U[i]); Rc[i] = opaque(R[i]); }
            // word by word, most significant first: two temporaries instead of two extracted windows (the hash loop has no
            // registers to spare, and a spill here gives the whole kernel a scratch allocation)
            constexpr uint32_t last_mask = (K % 4) ? (1u << (8 * (K % 4))) - 1u : 0xFFFFFFFFu;
            bool rc_less = false, decided = false;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                uint32_t f = funnel(Uc[OF / 4 + i + 1], Uc[OF / 4 + i], OF % 4);
                uint32_t r = funnel(Rc[OR / 4 + i + 1], Rc[OR / 4 + i], OR % 4);
                if (i == NW - 1) { f &= last_mask; r &= last_mask; }
                const uint32_t fb = opaque(__builtin_bswap32(f)), rb = opaque(__builtin_bswap32(r));
                rc_less = decided ? rc_less : (rb < fb);
                decided = decided || (fb != rb);
            }
            rc = rc_less;
        }
        // select the source dwords first, extract once
        uint32_t S[NW + 1];
#pragma unroll
        for (int i = 0; i < NW + 1; ++i) S[i] = rc ? R[OR / 4 + i] : U[OF / 4 + i];
        const uint32_t sh = rc ? 8u * (OR % 4) : 8u * (OF % 4);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            w[i] = i < NW ? funnel_bits(S[i + 1], S[i], sh) : 0u;
        if (K % 4) w[NW - 1] &= (1u << (8 * (K % 4))) - 1u;
    } else {
        uint32_t wf[8], wr[8];
        extract_words<K, OF>(U, wf);
        extract_words<K, OR>(R, wr);
        const bool rc = rc_is_smaller_full<NW>(wf, wr);
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = rc ? wr[i] : wf[i];
    }
}

// P5: one work item = 8 consecutive window starts sharing one register chunk.
template <int K> struct GroupGeom {
    static constexpr int NB = kGroup + K - 1; // bytes touched
    static constexpr int ND = (NB + 3) / 4;   // dwords loaded
};

// Cold path, reached by one window in ~2^64 / T: are bytes J .. J+K-1 of the chunk all A/C/G/T (either case)?
// mash skips every window that holds anything else (Sketch.cpp addMinHashes).
// (chunk: the raw or the case-folded bytes, the test folds anyway)
template <int K, int J, int ND> MHX_HD bool window_is_acgt(const uint32_t (&chunk)[ND + 1])
{
    uint64_t m = 0;
#pragma unroll
    for (int d = 0; d < ND; ++d) m |= (uint64_t)flags_to_nibble(acgt_flags(chunk[d])) << (4 * d);
    constexpr uint64_t want = (K >= 64 ? ~0ull : ((1ull << K) - 1ull)) << J;
    return (m & want) == want;
}

// vm: bits 0..7 the candidate-start mask of the group, bits 8.. the group's number (it rides along in the same register:
// the hash loop has none to spare).  A window that is a valid start and whose partial hash passes the admission test
// (`limit` = admission_limit(T); 32-bit hashes: the exact low word against T) is
//   QUEUE = false: finished where it is found: ins(h) if its bases are all A/C/G/T and h <= T (returns their number);
//   QUEUE = true : handed to cand(group, J) (queued, finished later by process_deferred).
// Two forms because each costs the other's workload something: with a small sketch (s = 1000, one candidate in 10^4
// windows) the inline form is 0.8 % faster (the queue's cold branch changes the register allocation of the hot path by
// one v_mov per window); with a large one (s = 50 000) the queue is 7 % faster.
template <int K, bool QUEUE, class Ins, class Cand>
MHX_HD uint32_t process_group_regs(const uint32_t (&src)[GroupGeom<K>::ND], uint32_t vm, uint64_t T, uint32_t limit, Ins &ins, Cand &cand)
{
    constexpr int ND = GroupGeom<K>::ND;
    uint32_t U[ND + 1], R[ND + 1], Wr[ND + 1], Cc[ND + 1];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        U[d] = src[d] & 0xDFDFDFDFu; // fold case: mash upper-cases before hashing
        Cc[d] = complement4(U[d]);
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        R[d] = __builtin_bswap32(Cc[ND - 1 - d]);
        Wr[d] = __builtin_bswap32(U[ND - 1 - d]);
    }
    U[ND] = R[ND] = Wr[ND] = Cc[ND] = 0;
    constexpr bool kHash32 = K <= 16; // mash keeps 32 bits when 4^k <= 2^32
    uint32_t ninserted = 0;
#define MHX_WINDOW(J)                                                                         \
    {                                                                                         \
        uint32_t w[8];                                                                        \
        canonical_words<K, J, ND>(U, R, Wr, Cc, w);                                           \
        const Murmur3Tail tail = murmur3_core<K>(w);                                          \
        /* necessary condition of h <= T, one add + one compare (32-bit hashes: the exact low word) */ \
        const bool candidate = kHash32 ? tail.low32() <= (uint32_t)T : tail.high_bound() <= limit; \
        if (MHX_UNLIKELY(candidate)) {                                                        \
            if constexpr (QUEUE) {                                                            \
                if ((vm >> J) & 1u) cand(vm >> 8, J);                                         \
            } else {                                                                          \
                const uint64_t h = kHash32 ? (uint64_t)tail.low32() : tail.finish();          \
                if (((vm >> J) & 1u) && h <= T && window_is_acgt<K, J, ND>(U)) { ins(h); ++ninserted; } \
            }                                                                                 \
        }                                                                                     \
    }
    MHX_WINDOW(0) MHX_WINDOW(1) MHX_WINDOW(2) MHX_WINDOW(3) MHX_WINDOW(4) MHX_WINDOW(5) MHX_WINDOW(6) MHX_WINDOW(7)
#undef MHX_WINDOW
    static_assert(kGroup == 8, "process_group unrolls 8 windows");
    return ninserted;
}

// work item g of a staged tile (LDS source)
template <int K, bool QUEUE, class Ins, class Cand>
MHX_HD uint32_t process_group(const TileSmem &sm, uint32_t g, uint64_t T, uint32_t limit, Ins &ins, Cand &cand)
{
    constexpr int ND = GroupGeom<K>::ND;
    const uint32_t vm = reinterpret_cast<const uint8_t *>(sm.valid)[g] | (g << 8);
    const uint32_t *p = reinterpret_cast<const uint32_t *>(sm.bytes) + 2 * g;
    uint32_t src[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) src[d] = p[d];
    return process_group_regs<K, QUEUE>(src, vm, T, limit, ins, cand);
}

// A queued candidate (code = (group << 3) | window), finished from the staged bytes with a window offset known only at
// run time: the K bytes by funnel shifts out of LDS, base check, reverse complement of those K bytes, the exact strand
// comparison, the whole hash.  Slower per window than the unrolled hot path, but run one candidate per lane.
template <int K, class Ins> MHX_HD uint32_t process_deferred(const TileSmem &sm, uint32_t code, uint64_t T, Ins &ins)
{
    constexpr int NW = (K + 3) / 4;
    constexpr uint32_t tail_mask = (K % 4) ? (1u << (8 * (K % 4))) - 1u : 0xFFFFFFFFu;
    const uint32_t pos = 8u * (code >> 3) + (code & 7u);
    const uint32_t *b = reinterpret_cast<const uint32_t *>(sm.bytes) + (pos >> 2);
    const uint32_t sh = 8u * (pos & 3u);
    uint32_t wf[8], wr[8], rev[NW + 1];
    uint32_t ok_bits = 0;
#pragma unroll
    for (int d = 0; d < NW; ++d) {
        const uint32_t raw = funnel_bits(b[d + 1], b[d], sh);
        ok_bits |= flags_to_nibble(acgt_flags(raw)) << (4 * d);
        wf[d] = raw & 0xDFDFDFDFu; // fold case
    }
    constexpr uint32_t want = (uint32_t)((1ull << K) - 1ull);
    if ((ok_bits & want) != want) return 0u; // mash skips every window that holds anything but A/C/G/T
    wf[NW - 1] &= tail_mask;
#pragma unroll
    for (int d = NW; d < 8; ++d) wf[d] = 0u;
    // reverse complement: the 4*NW bytes reversed and complemented put the window's K bytes behind 4*NW - K bytes of padding
#pragma unroll
    for (int d = 0; d < NW; ++d) rev[d] = __builtin_bswap32(complement4(wf[NW - 1 - d]));
    rev[NW] = 0u;
    constexpr uint32_t pad_bits = 8u * (4 * NW - K);
#pragma unroll
    for (int d = 0; d < 8; ++d) wr[d] = d < NW ? (pad_bits ? funnel_bits(rev[d + 1], rev[d], pad_bits) : rev[d]) : 0u;
    const bool rc = rc_is_smaller_full<NW>(wf, wr);
    uint32_t w[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) w[d] = rc ? wr[d] : wf[d];
    const Murmur3Tail tail = murmur3_core<K>(w);
    const uint64_t h = K <= 16 ? (uint64_t)tail.low32() : tail.finish();
    if (h > T) return 0u;
    ins(h);
    return 1u;
}

} // namespace mhx
