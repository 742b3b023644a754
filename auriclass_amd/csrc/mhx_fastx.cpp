// mhx_fastx.cpp -- host-side ingest: (gz) file -> bytes, and the record reader used when a
// stream cannot go to the device parser as it is (FASTA line unwrapping, multi-line FASTQ).
// Mirrors what `mash sketch` does with zlib + kseq.h before its hot loop (Mash 2.x
// Sketch.cpp sketchFile); the files are the ones AuriClass passes through unchanged at
// /root/reference/auriclass/classes.py:588 and :705.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>

#include <exception>
#include <new>
#include <thread>

#include "mhx_internal.h"

namespace mhx {

// whole file -> bytes; a gzip file (any number of members) is inflated with the engine's own decoder
// (mhx_inflate.cpp; MHX_ZLIB_INFLATE=1: zlib's gzread, which is also what handles non-regular inputs)
static int read_all_zlib(const char *path, std::vector<uint8_t> &out)
{
    gzFile g = gzopen(path, "rb"); // transparent for uncompressed files, like mash
    if (!g) return fail(MHX_E_IO, "ERROR: could not open %s for reading", path);
    gzbuffer(g, 1 << 20);
    out.clear();
    size_t cap = 1 << 22;
    out.resize(cap);
    size_t n = 0;
    for (;;) {
        if (n == cap) { cap *= 2; out.resize(cap); }
        const size_t want = cap - n > (1u << 30) ? (1u << 30) : cap - n;
        const int got = gzread(g, out.data() + n, (unsigned)want);
        if (got < 0) { gzclose(g); return fail(MHX_E_IO, "ERROR: reading %s failed", path); }
        if (got == 0) break;
        n += (size_t)got;
    }
    gzclose(g);
    out.resize(n);
    return MHX_OK;
}

int read_all_maybe_gz(const char *path, std::vector<uint8_t> &out)
{
    struct stat sb;
    if (getenv("MHX_ZLIB_INFLATE") || stat(path, &sb) != 0 || !S_ISREG(sb.st_mode)) return read_all_zlib(path, out);
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MHX_E_IO, "ERROR: could not open %s for reading", path);
    std::vector<uint8_t> raw((size_t)sb.st_size + GzInflater::kInputPad, 0);
    const size_t got = fread(raw.data(), 1, (size_t)sb.st_size, f);
    fclose(f);
    if (got != (size_t)sb.st_size) return fail(MHX_E_IO, "ERROR: reading %s failed", path);
    if (got < 18 || raw[0] != 0x1f || raw[1] != 0x8b) { // not gzip: the bytes as they are
        raw.resize(got);
        out.swap(raw);
        return MHX_OK;
    }
    GzInflater inf;
    inf.set_input(raw.data(), got);
    // size hint: ISIZE of the last member -- untrusted, so never more than 64x the compressed size up front
    // (DEFLATE text rarely passes 10x; the doubling below covers whatever the hint missed)
    size_t hint = (size_t)raw[got - 4] | ((size_t)raw[got - 3] << 8) | ((size_t)raw[got - 2] << 16) | ((size_t)raw[got - 1] << 24);
    hint = std::min<size_t>(hint, got * 64);
    const size_t slack = GzInflater::kOvershoot + 16;
    out.assign(std::max<size_t>(hint, 1u << 16) + slack, 0);
    size_t n = 0;
    const int threads = ingest_thread_budget();
    {   // bgzip output (assemblies kept indexable with faidx): the blocks side by side; what follows them, sequentially
        BgzfReader bgzf;
        if (bgzf.start(raw.data(), got, threads)) {
            for (;;) {
                if (out.size() - slack - n < (1u << 16)) out.resize(out.size() * 2);
                const size_t r = bgzf.read(out.data() + n, out.size() - slack - n);
                if (r == (size_t)-1) return read_all_zlib(path, out);
                if (r == 0) break;
                n += r;
            }
            const size_t off = bgzf.consumed_input();
            if (off >= got) { out.resize(n); return MHX_OK; }
            inf.set_input(raw.data() + off, got - off);
        }
    }
    if (n == 0) { // one ordinary member of a few MB (a gzipped assembly): several threads, segments of 256 KiB
        ParallelGunzip par;
        // (a thread per 256 KiB of compressed bytes at most: every segment starts with a search for a block boundary, and
        // a 3.5 MB assembly cut into 64 pieces for 32 threads took 8.6 ms where 14 threads take 5)
        const int member_threads = (int)std::min<size_t>((size_t)threads, std::max<size_t>(2, got / (256u << 10)));
        if (par.start(raw.data(), got, member_threads, 1u << 20, 256u << 10)) {
            bool ok = true;
            for (;;) {
                if (out.size() - slack - n < (1u << 16)) out.resize(out.size() * 2);
                const size_t r = par.read(out.data() + n, out.size() - slack - n);
                if (r == (size_t)-1) { ok = false; break; }
                if (r == 0) break;
                n += r;
            }
            if (ok) {
                const size_t off = par.consumed_input();
                if (off >= got) { out.resize(n); return MHX_OK; }
                inf.set_input(raw.data() + off, got - off);
            } else {
                n = 0; // declined or refused: the sequential decoder (and zlib behind it) has the last word
            }
        }
    }
    for (;;) {
        // (the decoder may run up to kOvershoot bytes past the room it is given -- `slack` is there for that --, so `n` can
        // end beyond size - slack: the room left is computed without wrapping.  It did wrap: a file of several members
        // whose last one announces less than the whole -- `cat a.gz b.gz` -- spun here for ever, asking for 2^64 bytes)
        const size_t room = out.size() - slack > n ? out.size() - slack - n : 0;
        if (room < (1u << 16)) { out.resize(out.size() * 2); continue; } // more members than the hint covered
        const size_t r = inf.inflate(out.data() + n, room, out.data());
        if (r == (size_t)-1) return read_all_zlib(path, out); // zlib has the last word on a stream this decoder refuses
        n += r;
        if (inf.done()) break;
    }
    if (n > out.size()) return fail(MHX_E_INTERNAL, "inflate wrote past its buffer"); // (cannot happen: room + kOvershoot < slack + room)
    out.resize(n);
    return MHX_OK;
}

int ingest_thread_budget()
{
    if (const char *e = getenv("MHX_INGEST_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
    long n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (long)std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2 quota: "max 100000" or "<quota> <period>"
        long long quota = 0, period = 0;
        // (twice the quota: the decoding threads spend a good part of their time waiting for each other's windows --
        // on the GPU box, 16 CPUs of quota, 32 threads beat 16 by 14 %)
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) n = std::min<long>(n, 2 * (long)((quota + period - 1) / period));
        fclose(f);
    }
    long ranks = 1;
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1L, atol(e));
    n /= ranks;
    return (int)std::min(32L, std::max(2L, n)); // measured on a 256-core host: 32 decode threads beat 16 and 64 (profiles/r03_c3_inclusive.txt)
}

static inline bool is_space(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

void first_header(const uint8_t *buf, size_t n, std::string &name, std::string &comment)
{
    name.clear();
    comment.clear();
    size_t p = 0;
    while (p < n && buf[p] != '>' && buf[p] != '@') ++p;
    if (p >= n) return;
    size_t e = ++p;
    while (e < n && buf[e] != '\n') ++e;
    size_t end = e;
    if (end > p && buf[end - 1] == '\r') --end;
    size_t sp = p;
    while (sp < end && buf[sp] != ' ' && buf[sp] != '\t') ++sp;
    name.assign(reinterpret_cast<const char *>(buf + p), sp - p);
    if (sp < end) comment.assign(reinterpret_cast<const char *>(buf + sp + 1), end - sp - 1);
}

// Cheap host-side plausibility test for the device FASTQ parser: starts with '@', and the
// first record has the 4-line shape.  The device verifies every record (kFlagBadFastq).
// Is the LAST record of a 4-line FASTQ complete in kseq's sense?  t: the last bytes of the stream.  A record that has its
// '+' line but no quality string, or one of another length than its sequence, makes kseq_read return -2 (the oracle
// raises "truncated quality string"): the device parser, which checks the four-line layout but not the quality lengths,
// must not sketch such a file -- the caller hands it to the host record parser, which reports it.  A header without
// anything behind it, or a sequence line without a '+' line, is a record to kseq (it reads it as FASTA).  true when the
// record is complete or the tail cannot tell (no record start among its last four lines).
bool fastq_tail_complete(const uint8_t *t, size_t n)
{
    size_t ls[5], le[5]; // the last (up to five) lines, newest first: [start, end) without the newline
    int nl = 0;
    size_t e = n;
    if (e && t[e - 1] == '\n') --e;                 // the final newline ends the last line
    else if (e == 0) return true;
    while (nl < 5) {
        size_t b = e;
        while (b > 0 && t[b - 1] != '\n') --b;
        ls[nl] = b; le[nl] = e; ++nl;
        if (b == 0) break;
        e = b - 1;
    }
    auto first = [&](int i) { return ls[i] < le[i] ? t[ls[i]] : (uint8_t)0; };
    auto length = [&](int i) { size_t l = le[i] - ls[i]; if (l && t[le[i] - 1] == '\r') --l; return l; };
    // the record start nearest to the end: a line that begins with '@', whose successor (if any) begins with neither '@'
    // nor '+', and whose second successor (if any) begins with '+'  (index 0 = last line; successors have smaller indices)
    for (int i = 0; i < nl && i < 4; ++i) {
        if (first(i) != '@') continue;
        if (i >= 1 && (first(i - 1) == '@' || first(i - 1) == '+')) continue;
        if (i >= 2 && first(i - 2) != '+') continue;
        if (i <= 1) return true;                                  // header only, or header + sequence: a record without qualities
        if (i == 2) return length(1) == 0;                        // '+' line, nothing behind it
        return length(0) == length(2);                            // i == 3: quality against sequence
    }
    return true;
}

bool looks_like_fastq4(const uint8_t *buf, size_t n)
{
    if (n == 0 || buf[0] != '@') return false;
    const uint8_t *l1 = (const uint8_t *)memchr(buf, '\n', n);
    if (!l1) return false;
    const uint8_t *l2 = (const uint8_t *)memchr(l1 + 1, '\n', n - (l1 + 1 - buf));
    if (!l2 || l2 + 1 >= buf + n) return false;
    if (l2[1] != '+') return false;
    const size_t look = n < (1u << 16) ? n : (1u << 16);
    return fastq_tail_complete(buf + n - look, look); // a last record cut short goes to the record parser, which reports it
}

// Record reader with kseq.h semantics: a record starts at '>' or '@'; name = header up to
// the first blank, comment = rest of the line; sequence = following lines (blanks removed)
// up to a line starting with '>', '@' or '+'; after '+' the quality is consumed until it is
// as long as the sequence.  Records shorter than k are dropped before they count.
int parse_fastx(const uint8_t *buf, size_t n, int k, ParsedRecords &out)
{
    size_t p = 0;
    while (p < n && buf[p] != '>' && buf[p] != '@') ++p;
    out.seq.reserve(out.seq.size() + (n - p) / 2 + 16);
    while (p < n) {
        // header
        size_t e = p + 1;
        {
            const uint8_t *nl = e < n ? (const uint8_t *)memchr(buf + e, '\n', n - e) : nullptr;
            e = nl ? (size_t)(nl - buf) : n;
        }
        const size_t hdr_begin = p + 1;
        size_t hdr_end = e;
        if (hdr_end > hdr_begin && buf[hdr_end - 1] == '\r') --hdr_end;
        p = e < n ? e + 1 : n;
        // sequence lines
        const size_t seq_start = out.seq.size();
        while (p < n && buf[p] != '>' && buf[p] != '@' && buf[p] != '+') {
            const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', n - p);
            const size_t le = nl ? (size_t)(nl - buf) : n;
            // append the line in bulk; blanks / control bytes (rare) are squeezed out afterwards
            const size_t at = out.seq.size();
            out.seq.insert(out.seq.end(), buf + p, buf + le);
            unsigned dirty = 0;
            for (size_t q = p; q < le; ++q) dirty |= (unsigned)(buf[q] <= ' ') | (unsigned)(buf[q] == 127);
            if (dirty) {
                size_t w = at;
                for (size_t q = at; q < out.seq.size(); ++q) {
                    const uint8_t c = out.seq[q];
                    if (c > ' ' && c != 127) out.seq[w++] = c;
                }
                out.seq.resize(w);
            }
            p = le < n ? le + 1 : n;
        }
        const size_t len = out.seq.size() - seq_start;
        if (p < n && buf[p] == '+') {
            while (p < n && buf[p] != '\n') ++p;
            if (p < n) ++p;
            size_t ql = 0;
            while (p < n && ql < len) {
                size_t le = p;
                while (le < n && buf[le] != '\n') ++le;
                for (size_t q = p; q < le; ++q) if (buf[q] > ' ' && buf[q] != 127) ++ql;
                p = le < n ? le + 1 : n;
            }
            if (ql != len) return fail(MHX_E_FORMAT, "truncated quality string in FASTQ record %llu", (unsigned long long)out.records_seen + 1);
            while (p < n && buf[p] != '>' && buf[p] != '@') ++p;
        }
        ++out.records_seen;
        if (len < (size_t)k) {
            out.skipped_short = true;
            out.seq.resize(seq_start);
            continue;
        }
        if (out.records == 0) {
            size_t sp = hdr_begin;
            while (sp < hdr_end && buf[sp] != ' ' && buf[sp] != '\t') ++sp;
            out.first_name.assign(reinterpret_cast<const char *>(buf + hdr_begin), sp - hdr_begin);
            out.first_comment.clear();
            if (sp < hdr_end) out.first_comment.assign(reinterpret_cast<const char *>(buf + sp + 1), hdr_end - sp - 1);
        }
        ++out.records;
        out.total_length += len;
        out.seq.push_back('\n'); // record separator: no k-mer spans two records
    }
    return MHX_OK;
}

} // namespace mhx

// ---- C ABI: sniffers and FASTA size (replace the pyfastx calls of the reference) ----------
using namespace mhx;

// pyfastx.Fastq / pyfastx.Fasta accept a file when its first record parses in that format
// (/root/reference/auriclass/general.py:68-115).  Same decision from the first bytes.
static int sniff(const char *path, char lead)
{
    gzFile g = gzopen(path, "rb");
    if (!g) return fail(MHX_E_IO, "cannot open %s", path);
    uint8_t buf[1 << 16];
    const int got = gzread(g, buf, sizeof(buf));
    gzclose(g);
    if (got < 0) return 0; // unreadable as text: neither format
    int p = 0;
    while (p < got && is_space(buf[p])) ++p;
    if (p >= got || buf[p] != (uint8_t)lead) return 0;
    // one header line followed by at least one sequence character
    const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', got - p);
    if (!nl || nl + 1 >= buf + got) return 0;
    const uint8_t c = nl[1];
    if (!((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '*' || c == '-')) return 0;
    if (lead == '@') {
        const uint8_t *nl2 = (const uint8_t *)memchr(nl + 1, '\n', got - (nl + 1 - buf));
        if (!nl2 || nl2 + 1 >= buf + got || nl2[1] != '+') return 0;
    }
    return 1;
}

static int sniff_guarded(const char *path, char lead)
{
    clear_error();
    if (!path) return fail(MHX_E_ARG, "null path");
    try {
        return sniff(path, lead);
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "sniff: %s", e.what());
    }
}
extern "C" int mhx_sniff_fastq(const char *path) { return sniff_guarded(path, '@'); }
extern "C" int mhx_sniff_fasta(const char *path) { return sniff_guarded(path, '>'); }
extern "C" int mhx_fastq_tail_complete(const void *tail, size_t n) { return tail || n == 0 ? (fastq_tail_complete((const uint8_t *)tail, n) ? 1 : 0) : 1; }

static int fasta_total_bases_impl(const char *path, uint64_t *total)
{
    clear_error();
    if (!path || !total) return fail(MHX_E_ARG, "null argument");
    std::vector<uint8_t> raw;
    int rc = read_all_maybe_gz(path, raw);
    if (rc) return rc;
    ParsedRecords pr;
    rc = parse_fastx(raw.data(), raw.size(), 0, pr);
    if (rc) return rc;
    *total = pr.total_length;
    return MHX_OK;
}

// no exception may cross the C boundary (ctypes would abort the process): a crafted .gz can ask for gigabytes
extern "C" int mhx_fasta_total_bases(const char *path, uint64_t *total)
{
    try {
        return fasta_total_bases_impl(path, total);
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_fasta_total_bases: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_fasta_total_bases: %s", e.what());
    }
}
