// mhx_msh.cpp -- reader/writer of mash's `.msh` sketch container: an UNPACKED Cap'n Proto
// message of schema MinHash.capnp (Mash 2.x).  No capnp library is used; the byte layout
// -- including the arena behaviour of capnp's MallocMessageBuilder, which decides how the
// message is split into segments -- is reproduced so that the files are byte-identical
// to the ones `mash sketch` writes (golden: /root/reference/tests/data/ref_sketch.msh,
// consumed by /root/reference/auriclass/classes.py:92-97 via `mash dist`).
//
// Message layout (words of 8 bytes):
//   root struct  3 data words | 4 pointers
//     data  kmerSize u32@0, windowSize u32@4, minHashesPerWindow u32@8, flag bits @96..98
//           (concatenated, noncanonical, preserveCase), error f32@16, hashSeed^42 u32@20
//     ptrs  [0] referenceListOld (used when seed == 42)  [1] locusList  [2] alphabet  [3] referenceList
//   ReferenceList { references: List(Reference) }, Reference = 2 data words | 7 pointers
//     data  length u32@0 (unused), length64 u64@8
//     ptrs  sequence, quality, name, comment, hashes32, hashes64, counts32
#include <stdio.h>
#include <string.h>

#include "mhx_internal.h"

namespace mhx {
namespace {

// -- arena: segment 0 holds 1024 words; every later segment is max(request, words allocated
// so far).  An object goes to its pointer's segment when it fits; otherwise it is placed
// (behind a one-word landing pad, reached by a far pointer) in the newest segment, or in a
// new segment when that one is full as well.
class Arena {
  public:
    struct Place { uint32_t seg; uint32_t word; bool far; };
    std::vector<std::vector<uint64_t>> segs;
    std::vector<uint32_t> caps;

    Place root()
    {
        add_segment(1);
        return {0, take(0, 1), false};
    }
    Place alloc_for(uint32_t ptr_seg, uint32_t words)
    {
        if (fits(ptr_seg, words)) return {ptr_seg, take(ptr_seg, words), false};
        const uint32_t last = (uint32_t)segs.size() - 1;
        if (last != ptr_seg && fits(last, words + 1)) return {last, take(last, words + 1), true};
        const uint32_t s = add_segment(words + 1);
        return {s, take(s, words + 1), true};
    }
    uint64_t &at(uint32_t seg, uint32_t word) { return segs[seg][word]; }
    uint8_t *bytes(uint32_t seg, uint32_t word) { return reinterpret_cast<uint8_t *>(&segs[seg][word]); }

  private:
    uint64_t next_size = 1024;
    bool fits(uint32_t seg, uint32_t words) const { return segs[seg].size() + words <= caps[seg]; }
    uint32_t take(uint32_t seg, uint32_t words)
    {
        const uint32_t at = (uint32_t)segs[seg].size();
        segs[seg].resize(at + words, 0);
        return at;
    }
    uint32_t add_segment(uint64_t min_words)
    {
        const uint64_t size = min_words > next_size ? min_words : next_size;
        if (segs.empty()) next_size = size; else next_size += size;
        segs.emplace_back();
        segs.back().reserve(size < (1u << 20) ? size : min_words);
        caps.push_back((uint32_t)size);
        return (uint32_t)segs.size() - 1;
    }
};

inline uint64_t struct_ptr(int32_t off, uint32_t dw, uint32_t np) { return ((uint64_t)((uint32_t)off & 0x3FFFFFFFu) << 2) | ((uint64_t)dw << 32) | ((uint64_t)np << 48); }
inline uint64_t list_ptr(int32_t off, uint32_t code, uint64_t count) { return 1ull | ((uint64_t)((uint32_t)off & 0x3FFFFFFFu) << 2) | ((uint64_t)code << 32) | (count << 35); }
inline uint64_t far_ptr(uint32_t seg, uint32_t pad) { return 2ull | ((uint64_t)pad << 3) | ((uint64_t)seg << 32); }

struct Builder {
    Arena a;
    enum Kind { STRUCT, LIST };
    // allocate `words` for the pointer at (pseg,pword); encode it near or far; returns object position
    Arena::Place place(uint32_t pseg, uint32_t pword, uint32_t words, Kind kind, uint32_t x, uint64_t y)
    {
        Arena::Place p = a.alloc_for(pseg, words);
        auto enc = [&](int32_t off) { return kind == STRUCT ? struct_ptr(off, x, (uint32_t)y) : list_ptr(off, x, y); };
        if (!p.far) {
            a.at(pseg, pword) = enc((int32_t)p.word - (int32_t)(pword + 1));
            return p;
        }
        a.at(pseg, pword) = far_ptr(p.seg, p.word);
        a.at(p.seg, p.word) = enc(0);
        return {p.seg, p.word + 1, true};
    }
    Arena::Place new_struct(uint32_t pseg, uint32_t pword, uint32_t dw, uint32_t np) { return place(pseg, pword, dw + np, STRUCT, dw, np); }
    void text(uint32_t pseg, uint32_t pword, const std::string &s)
    {
        const uint64_t n = s.size() + 1;
        Arena::Place p = place(pseg, pword, (uint32_t)((n + 7) / 8), LIST, 2, n);
        memcpy(a.bytes(p.seg, p.word), s.data(), s.size());
    }
    void prim_list(uint32_t pseg, uint32_t pword, const void *data, uint64_t count, uint32_t elem_bytes)
    {
        const uint64_t nb = count * elem_bytes;
        Arena::Place p = place(pseg, pword, (uint32_t)((nb + 7) / 8), LIST, elem_bytes == 8 ? 5 : 4, count);
        memcpy(a.bytes(p.seg, p.word), data, nb);
    }
    Arena::Place composite(uint32_t pseg, uint32_t pword, uint32_t count, uint32_t dw, uint32_t np)
    {
        const uint64_t words = (uint64_t)count * (dw + np);
        Arena::Place p = place(pseg, pword, (uint32_t)words + 1, LIST, 7, words);
        a.at(p.seg, p.word) = ((uint64_t)(count & 0x3FFFFFFFu) << 2) | ((uint64_t)dw << 32) | ((uint64_t)np << 48);
        return {p.seg, p.word + 1, p.far};
    }
};

} // namespace

// Same allocation order as mash Sketch::writeToCapnp: root, reference list, then per
// reference name / comment / hashes (/counts), locus list, and the alphabet last.
int msh_serialize(const SketchSet &s, std::vector<uint8_t> &out)
{
    Builder b;
    Arena::Place r = b.a.root();
    Arena::Place root = b.new_struct(r.seg, r.word, 3, 4);
    const uint32_t pbase = root.word + 3;
    Arena::Place rl = b.new_struct(root.seg, pbase + (s.hash_seed == 42 ? 0 : 3), 0, 1);
    Arena::Place el = b.composite(rl.seg, rl.word, (uint32_t)s.refs.size(), 2, 7);
    std::vector<uint32_t> tmp32;
    for (size_t i = 0; i < s.refs.size(); ++i) {
        const RefSketch &ref = s.refs[i];
        const uint32_t e = el.word + (uint32_t)i * 9;
        b.text(el.seg, e + 2 + 2, ref.name);
        b.text(el.seg, e + 2 + 3, ref.comment);
        b.a.at(el.seg, e + 1) = ref.length;
        if (!ref.hashes.empty()) {
            if (s.use64()) {
                b.prim_list(el.seg, e + 2 + 5, ref.hashes.data(), ref.hashes.size(), 8);
            } else {
                tmp32.resize(ref.hashes.size());
                for (size_t j = 0; j < tmp32.size(); ++j) tmp32[j] = (uint32_t)ref.hashes[j];
                b.prim_list(el.seg, e + 2 + 4, tmp32.data(), tmp32.size(), 4);
            }
            if (!ref.counts.empty()) b.prim_list(el.seg, e + 2 + 6, ref.counts.data(), ref.counts.size(), 4);
        }
    }
    Arena::Place ll = b.new_struct(root.seg, pbase + 1, 0, 1);
    b.composite(ll.seg, ll.word, 0, 3, 0);
    const uint64_t flags = (s.concatenated ? 1u : 0u) | (s.noncanonical ? 2u : 0u) | (s.preserve_case ? 4u : 0u);
    uint32_t err_bits;
    memcpy(&err_bits, &s.error, 4);
    b.a.at(root.seg, root.word + 0) = (uint64_t)s.kmer_size | ((uint64_t)s.window_size << 32);
    b.a.at(root.seg, root.word + 1) = (uint64_t)s.sketch_size | (flags << 32);
    b.a.at(root.seg, root.word + 2) = (uint64_t)err_bits | ((uint64_t)(s.hash_seed ^ 42u) << 32);
    b.text(root.seg, pbase + 2, s.alphabet);

    const uint32_t nseg = (uint32_t)b.a.segs.size();
    size_t hdr = 4 + 4 * (size_t)nseg;
    if (hdr % 8) hdr += 4;
    size_t total = hdr;
    for (auto &sg : b.a.segs) total += sg.size() * 8;
    out.assign(total, 0);
    uint32_t v = nseg - 1;
    memcpy(&out[0], &v, 4);
    for (uint32_t i = 0; i < nseg; ++i) { v = (uint32_t)b.a.segs[i].size(); memcpy(&out[4 + 4 * i], &v, 4); }
    size_t off = hdr;
    for (auto &sg : b.a.segs) { memcpy(&out[off], sg.data(), sg.size() * 8); off += sg.size() * 8; }
    return MHX_OK;
}

int msh_write_file(const char *path, const SketchSet &s)
{
    std::vector<uint8_t> buf;
    int rc = msh_serialize(s, buf);
    if (rc) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(MHX_E_IO, "cannot open %s for writing", path);
    const size_t w = fwrite(buf.data(), 1, buf.size(), f);
    if (fclose(f) != 0 || w != buf.size()) return fail(MHX_E_IO, "short write to %s", path);
    return MHX_OK;
}

// ---- reader ----------------------------------------------------------------------------
namespace {
struct Reader {
    std::vector<const uint64_t *> seg;
    std::vector<uint32_t> words;
    struct Target { bool ok; uint32_t kind, seg, word, hi; };
    bool in(uint32_t s, uint64_t w, uint64_t n = 1) const { return s < seg.size() && w + n <= words[s]; }
    Target resolve(uint32_t s, uint32_t w) const
    {
        if (!in(s, w)) return {false, 0, 0, 0, 0};
        uint64_t p = seg[s][w];
        if (p == 0) return {false, 0, 0, 0, 0};
        if ((p & 3) == 2) {
            if (p & 4) return {false, 0, 0, 0, 0}; // double-far: mash never writes it
            s = (uint32_t)(p >> 32);
            w = (uint32_t)((p >> 3) & 0x1FFFFFFF);
            if (!in(s, w)) return {false, 0, 0, 0, 0};
            p = seg[s][w];
        }
        int32_t off = (int32_t)((uint32_t)(p >> 2) & 0x3FFFFFFF);
        if (off & 0x20000000) off -= 0x40000000;
        const int64_t tgt = (int64_t)w + 1 + off;
        if (tgt < 0 || !in(s, (uint64_t)tgt, 0)) return {false, 0, 0, 0, 0};
        return {true, (uint32_t)(p & 3), s, (uint32_t)tgt, (uint32_t)(p >> 32)};
    }
    bool text(uint32_t s, uint32_t w, std::string &out) const
    {
        out.clear();
        Target t = resolve(s, w);
        if (!t.ok) return true;
        const uint32_t n = t.hi >> 3;
        if (n == 0 || !in(t.seg, t.word, (n + 7) / 8)) return n == 0;
        out.assign(reinterpret_cast<const char *>(seg[t.seg] + t.word), n - 1);
        return true;
    }
};
} // namespace

bool check_ascending(const uint64_t *h, size_t n)
{
    for (size_t j = 1; j < n; ++j)
        if (h[j] <= h[j - 1]) return false;
    return true;
}

int msh_read_file(const char *path, SketchSet &s)
{
    std::vector<uint8_t> raw;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MHX_E_IO, "cannot open sketch %s", path);
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    raw.resize(sz > 0 ? (size_t)sz : 0);
    if (sz > 0 && fread(raw.data(), 1, raw.size(), f) != raw.size()) { fclose(f); return fail(MHX_E_IO, "cannot read %s", path); }
    fclose(f);
    return msh_parse_image(raw.data(), raw.size(), path, s, false);
}

int msh_parse_image(const uint8_t *raw_data, size_t raw_size, const char *path, SketchSet &s, bool views)
{
    struct { const uint8_t *p; size_t n; const uint8_t *data() const { return p; } size_t size() const { return n; } } raw{raw_data, raw_size};
    if (raw.size() < 16) return fail(MHX_E_FORMAT, "%s is not a mash sketch (too short)", path);
    uint32_t nseg;
    memcpy(&nseg, raw.data(), 4);
    nseg += 1;
    if (nseg == 0 || nseg > 65536 || 4 + 4 * (size_t)nseg > raw.size()) return fail(MHX_E_FORMAT, "%s is not a mash sketch (segment table)", path);
    size_t off = 4 + 4 * (size_t)nseg;
    if (off % 8) off += 4;
    // segments are read as 8-byte words: where they are, if the image is 8-byte aligned (the segments start at a multiple
    // of 8 bytes from its start), else from copies
    const bool in_place = (reinterpret_cast<uintptr_t>(raw.data()) & 7u) == 0;
    if (!in_place) views = false;
    std::vector<std::vector<uint64_t>> store(in_place ? 0 : nseg);
    Reader rd;
    for (uint32_t i = 0; i < nseg; ++i) {
        uint32_t w;
        memcpy(&w, raw.data() + 4 + 4 * i, 4);
        if (off + (size_t)w * 8 > raw.size()) return fail(MHX_E_FORMAT, "%s is truncated", path);
        if (in_place) {
            rd.seg.push_back(reinterpret_cast<const uint64_t *>(raw.data() + off));
        } else {
            store[i].resize(w);
            if (w) memcpy(store[i].data(), raw.data() + off, (size_t)w * 8);
            rd.seg.push_back(store[i].data());
        }
        off += (size_t)w * 8;
        rd.words.push_back(w);
    }
    Reader::Target root = rd.resolve(0, 0);
    if (!root.ok || root.kind != 0) return fail(MHX_E_FORMAT, "%s: bad root pointer", path);
    const uint32_t dw = root.hi & 0xFFFF, np = root.hi >> 16;
    if (dw < 3 || np < 3 || !rd.in(root.seg, root.word, dw + np)) return fail(MHX_E_FORMAT, "%s: unexpected root struct", path);
    const uint64_t *d = rd.seg[root.seg] + root.word;
    s = SketchSet();
    s.kmer_size = (uint32_t)d[0];
    s.window_size = (uint32_t)(d[0] >> 32);
    s.sketch_size = (uint32_t)d[1];
    s.concatenated = (d[1] >> 32) & 1;
    s.noncanonical = (d[1] >> 33) & 1;
    s.preserve_case = (d[1] >> 34) & 1;
    const uint32_t eb = (uint32_t)d[2];
    memcpy(&s.error, &eb, 4);
    s.hash_seed = (uint32_t)(d[2] >> 32) ^ 42u;
    const uint32_t pbase = root.word + dw;
    rd.text(root.seg, pbase + 2, s.alphabet);
    Reader::Target rl = rd.resolve(root.seg, pbase + 0);
    if (!rl.ok && np > 3) rl = rd.resolve(root.seg, pbase + 3);
    if (rl.ok) {
        Reader::Target lst = rd.resolve(rl.seg, rl.word);
        if (lst.ok && lst.kind == 1 && (lst.hi & 7) == 7) {
            // resolve() accepts a target right behind the segment's last word (legal for an empty object): the tag
            // word of a composite list must itself be inside
            if (!rd.in(lst.seg, lst.word, 1)) return fail(MHX_E_FORMAT, "%s: bad reference list", path);
            const uint64_t tag = rd.seg[lst.seg][lst.word];
            const uint32_t count = (uint32_t)(tag >> 2) & 0x3FFFFFFF, ed = (uint32_t)(tag >> 32) & 0xFFFF, ep = (uint32_t)(tag >> 48);
            if (!rd.in(lst.seg, lst.word + 1, (uint64_t)count * (ed + ep)) || ed < 2 || ep < 6)
                return fail(MHX_E_FORMAT, "%s: bad reference list", path);
            s.refs.resize(count);
            for (uint32_t i = 0; i < count; ++i) {
                const uint32_t e = lst.word + 1 + i * (ed + ep), pp = e + ed;
                RefSketch &r = s.refs[i];
                r.length = rd.seg[lst.seg][e + 1] ? rd.seg[lst.seg][e + 1] : (uint32_t)rd.seg[lst.seg][e];
                rd.text(lst.seg, pp + 2, r.name);
                rd.text(lst.seg, pp + 3, r.comment);
                Reader::Target h64 = rd.resolve(lst.seg, pp + 5), h32 = rd.resolve(lst.seg, pp + 4);
                if (h64.ok) {
                    const uint32_t n = h64.hi >> 3;
                    if ((h64.hi & 7) != 5 || !rd.in(h64.seg, h64.word, n)) return fail(MHX_E_FORMAT, "%s: bad hash list", path);
                    if (views) { r.view = rd.seg[h64.seg] + h64.word; r.view_n = n; }
                    else r.hashes.assign(rd.seg[h64.seg] + h64.word, rd.seg[h64.seg] + h64.word + n);
                } else if (h32.ok) {
                    const uint32_t n = h32.hi >> 3;
                    if ((h32.hi & 7) != 4 || !rd.in(h32.seg, h32.word, ((uint64_t)n + 1) / 2)) return fail(MHX_E_FORMAT, "%s: bad hash list", path);
                    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(rd.seg[h32.seg] + h32.word);
                    r.hashes.resize(n);
                    for (uint32_t j = 0; j < n; ++j) r.hashes[j] = p32[j];
                }
                // the distance kernels merge ascending duplicate-free lists (what mash writes); anything else is a damaged file
                // (a list left in the image as a view is checked by the caller)
                if (!check_ascending(r.hashes.data(), r.hashes.size())) return fail(MHX_E_FORMAT, "%s: hash list of reference %u is not ascending", path, i);
                if (ep > 6) {
                    Reader::Target c = rd.resolve(lst.seg, pp + 6);
                    if (c.ok && (c.hi & 7) == 4) {
                        const uint32_t n = c.hi >> 3;
                        if (!rd.in(c.seg, c.word, ((uint64_t)n + 1) / 2)) return fail(MHX_E_FORMAT, "%s: bad count list", path);
                        const uint32_t *p32 = reinterpret_cast<const uint32_t *>(rd.seg[c.seg] + c.word);
                        r.counts.assign(p32, p32 + n);
                    }
                }
            }
        }
    }
    return MHX_OK;
}

} // namespace mhx
