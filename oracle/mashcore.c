/*
 * oracle/mashcore.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the arithmetic AuriClass obtains from the third-party
 * `mash` 2.x binary (bioconda mash=2, /root/reference/env.yaml:6), reached from
 * /root/reference/auriclass/classes.py:576-596 (`mash sketch -r -m M -k K -s S`),
 * classes.py:696-713 (`mash sketch -k K -s S`) and classes.py:92-104 (`mash dist`).
 * Mash's sources are not vendored in the reference, so this file restates the
 * published Mash 2.x algorithm (Sketch.cpp addMinHashes/getHash, MinHashHeap.cpp,
 * CommandDistance.cpp compareSketches, Austin Appleby's public-domain
 * MurmurHash3_x64_128) and is PINNED by the reference's own golden vectors
 * (tests/data/ref_sketch.msh, tests/test_correct_workflow.py:18-38,99,105,197);
 * see tests/test_oracle_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (auriclass_amd/) never links or calls it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------------- */
/* MurmurHash3_x64_128 (Appleby, public domain algorithm), as called by mash   */
/* getHash(): seed 42, first 8 output bytes kept (4 when 4^k <= 2^32).         */
/* ------------------------------------------------------------------------- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

void mo_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2])
{
    const uint8_t *data = (const uint8_t *)key;
    const int nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (int i = 0; i < nblocks; i++) {
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8);
        memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = data + nblocks * 16;
    uint64_t k1 = 0, k2 = 0;
    switch (len & 15) {
    case 15: k2 ^= ((uint64_t)tail[14]) << 48; /* fallthrough */
    case 14: k2 ^= ((uint64_t)tail[13]) << 40; /* fallthrough */
    case 13: k2 ^= ((uint64_t)tail[12]) << 32; /* fallthrough */
    case 12: k2 ^= ((uint64_t)tail[11]) << 24; /* fallthrough */
    case 11: k2 ^= ((uint64_t)tail[10]) << 16; /* fallthrough */
    case 10: k2 ^= ((uint64_t)tail[9]) << 8;   /* fallthrough */
    case 9:  k2 ^= ((uint64_t)tail[8]) << 0;
             k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
    case 8:  k1 ^= ((uint64_t)tail[7]) << 56; /* fallthrough */
    case 7:  k1 ^= ((uint64_t)tail[6]) << 48; /* fallthrough */
    case 6:  k1 ^= ((uint64_t)tail[5]) << 40; /* fallthrough */
    case 5:  k1 ^= ((uint64_t)tail[4]) << 32; /* fallthrough */
    case 4:  k1 ^= ((uint64_t)tail[3]) << 24; /* fallthrough */
    case 3:  k1 ^= ((uint64_t)tail[2]) << 16; /* fallthrough */
    case 2:  k1 ^= ((uint64_t)tail[1]) << 8;  /* fallthrough */
    case 1:  k1 ^= ((uint64_t)tail[0]) << 0;
             k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}

/* mash getHash(): hash64 when 4^k > 2^32 (k >= 17), else the low 32 bits. */
static inline int use64_for_k(int k) { return pow(4.0, (double)k) > pow(2.0, 32.0); }

uint64_t mo_kmer_hash(const char *kmer, int k, uint32_t seed)
{
    uint64_t out[2];
    mo_murmur3_x64_128(kmer, k, seed, out);
    return use64_for_k(k) ? out[0] : (out[0] & 0xffffffffULL);
}

/* ------------------------------------------------------------------------- */
/* (hash -> count) open-addressing map with backward-shift deletion, and a     */
/* binary max-heap: the two containers mash's MinHashHeap is built from.       */
/* ------------------------------------------------------------------------- */
typedef struct { uint64_t *keys; uint32_t *cnt; uint64_t cap, n; } map_t;

static inline uint64_t map_slot(const map_t *m, uint64_t h)
{
    return (h * 0x9E3779B97F4A7C15ULL) >> 7 & (m->cap - 1);
}
static void map_init(map_t *m, uint64_t cap)
{
    m->cap = cap; m->n = 0;
    m->keys = (uint64_t *)calloc(cap, sizeof(uint64_t));
    m->cnt = (uint32_t *)calloc(cap, sizeof(uint32_t));
}
static void map_free(map_t *m) { free(m->keys); free(m->cnt); }
static uint32_t map_count(const map_t *m, uint64_t h)
{
    uint64_t i = map_slot(m, h);
    while (m->cnt[i]) {
        if (m->keys[i] == h) return m->cnt[i];
        i = (i + 1) & (m->cap - 1);
    }
    return 0;
}
static void map_add(map_t *m, uint64_t h, uint32_t c);
static void map_grow(map_t *m)
{
    map_t n; map_init(&n, m->cap * 2);
    for (uint64_t i = 0; i < m->cap; i++) if (m->cnt[i]) map_add(&n, m->keys[i], m->cnt[i]);
    map_free(m); *m = n;
}
static void map_add(map_t *m, uint64_t h, uint32_t c)
{
    if ((m->n + 1) * 2 > m->cap) map_grow(m);
    uint64_t i = map_slot(m, h);
    while (m->cnt[i]) {
        if (m->keys[i] == h) { m->cnt[i] += c; return; }
        i = (i + 1) & (m->cap - 1);
    }
    m->keys[i] = h; m->cnt[i] = c; m->n++;
}
static void map_erase(map_t *m, uint64_t h)
{
    uint64_t mask = m->cap - 1, i = map_slot(m, h);
    while (m->cnt[i]) {
        if (m->keys[i] == h) break;
        i = (i + 1) & mask;
    }
    if (!m->cnt[i]) return;
    m->cnt[i] = 0; m->n--;
    uint64_t j = i;
    for (;;) {
        j = (j + 1) & mask;
        if (!m->cnt[j]) break;
        uint64_t home = map_slot(m, m->keys[j]);
        /* can entry j move into hole i?  yes iff home is cyclically outside (i, j] */
        int move = (i <= j) ? (home <= i || home > j) : (home <= i && home > j);
        if (move) { m->keys[i] = m->keys[j]; m->cnt[i] = m->cnt[j]; m->cnt[j] = 0; i = j; }
    }
}

typedef struct { uint64_t *a; uint64_t n, cap; } heap_t;
static void heap_push(heap_t *h, uint64_t v)
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 1024; h->a = (uint64_t *)realloc(h->a, h->cap * 8); }
    uint64_t i = h->n++;
    while (i) { uint64_t p = (i - 1) / 2; if (h->a[p] >= v) break; h->a[i] = h->a[p]; i = p; }
    h->a[i] = v;
}
static void heap_pop(heap_t *h)
{
    uint64_t v = h->a[--h->n], i = 0;
    for (;;) {
        uint64_t c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && h->a[c + 1] > h->a[c]) c++;
        if (h->a[c] <= v) break;
        h->a[i] = h->a[c]; i = c;
    }
    if (h->n) h->a[i] = v;
}

/* ------------------------------------------------------------------------- */
/* Sketch state == mash MinHashHeap (cardinalityMaximum = s,                   */
/* multiplicityMinimum = m) + the per-reference bookkeeping of sketchFile().   */
/* ------------------------------------------------------------------------- */
typedef struct mo_sketch {
    int k; uint32_t s, m; uint32_t seed; int use64;
    map_t hashes, pending;
    heap_t queue, queue_pending;
    uint64_t multiplicity_sum;
    uint64_t kmers_hashed;     /* valid windows fed to tryInsert */
    uint64_t records;          /* records with length >= k ("count" in sketchFile) */
    uint64_t length;           /* sum of those records' lengths */
    int skipped_short;
    char first_name[1024], first_comment[4096];
} mo_sketch;

mo_sketch *mo_sketch_new(int k, uint32_t s, uint32_t m)
{
    mo_sketch *sk = (mo_sketch *)calloc(1, sizeof(*sk));
    sk->k = k; sk->s = s; sk->m = m ? m : 1; sk->seed = 42; sk->use64 = use64_for_k(k);
    map_init(&sk->hashes, 1024); map_init(&sk->pending, 1024);
    return sk;
}
void mo_sketch_free(mo_sketch *sk)
{
    if (!sk) return;
    map_free(&sk->hashes); map_free(&sk->pending);
    free(sk->queue.a); free(sk->queue_pending.a); free(sk);
}

/* mash MinHashHeap::tryInsert */
static void try_insert(mo_sketch *sk, uint64_t h)
{
    if (sk->hashes.n < sk->s || h < sk->queue.a[0]) {
        if (map_count(&sk->hashes, h) == 0) {
            if (sk->m == 1 || map_count(&sk->pending, h) == sk->m - 1) {
                map_add(&sk->hashes, h, sk->m);
                heap_push(&sk->queue, h);
                sk->multiplicity_sum += sk->m;
                if (sk->m > 1) map_erase(&sk->pending, h);
            } else {
                if (map_count(&sk->pending, h) == 0) heap_push(&sk->queue_pending, h);
                map_add(&sk->pending, h, 1);
            }
        } else {
            map_add(&sk->hashes, h, 1);
            sk->multiplicity_sum++;
        }
        if (sk->hashes.n > sk->s) {
            uint64_t top = sk->queue.a[0];
            sk->multiplicity_sum -= map_count(&sk->hashes, top);
            map_erase(&sk->hashes, top);
            heap_pop(&sk->queue);
            /* drop pending hashes that can no longer enter (>= new top); zombies
               (already promoted) are simply popped */
            while (sk->queue_pending.n && sk->queue.n && sk->queue_pending.a[0] > sk->queue.a[0]) {
                map_erase(&sk->pending, sk->queue_pending.a[0]);
                heap_pop(&sk->queue_pending);
            }
        }
    }
}

/* mash addMinHashes(): upper-case in place, reverse complement, per-window
 * skip of non-ACGT, canonical = memcmp(fwd, rev) <= 0 ? fwd : rev, hash, insert.
 * `seq` is modified (upper-cased) exactly like mash does. */
void mo_sketch_add_seq(mo_sketch *sk, char *seq, uint64_t length)
{
    const int k = sk->k;
    if (length < (uint64_t)k) return;
    for (uint64_t i = 0; i < length; i++)
        if (seq[i] > 96 && seq[i] < 123) seq[i] -= 32;
    char *rev = (char *)malloc(length);
    for (uint64_t i = 0; i < length; i++) {
        char b = seq[i];
        switch (b) { case 'A': b = 'T'; break; case 'C': b = 'G'; break;
                     case 'G': b = 'C'; break; case 'T': b = 'A'; break; default: break; }
        rev[length - i - 1] = b;
    }
    uint64_t j = 0;
    for (uint64_t i = 0; i < length - k + 1; i++) {
        int bad = 0;
        for (; j < i + k && i + k <= length; j++) {
            char c = seq[j];
            if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) { i = j++; bad = 1; break; }
        }
        if (bad) continue;
        if (i + k > length) break;
        const char *fwd = seq + i, *rc = rev + length - i - k;
        const char *kmer = memcmp(fwd, rc, k) <= 0 ? fwd : rc;
        try_insert(sk, mo_kmer_hash(kmer, k, sk->seed));
        sk->kmers_hashed++;
    }
    free(rev);
}

/* One record as mash sketchFile() sees it after kseq_read(): l < k records are
 * skipped before they count; the first counted record provides the comment. */
static void add_record(mo_sketch *sk, const char *name, size_t name_len,
                       const char *comment, size_t comment_len, char *seq, uint64_t l)
{
    if (l < (uint64_t)sk->k) { sk->skipped_short = 1; return; }
    if (sk->records == 0) {
        if (name_len >= sizeof(sk->first_name)) name_len = sizeof(sk->first_name) - 1;
        if (comment_len >= sizeof(sk->first_comment)) comment_len = sizeof(sk->first_comment) - 1;
        memcpy(sk->first_name, name, name_len); sk->first_name[name_len] = 0;
        memcpy(sk->first_comment, comment, comment_len); sk->first_comment[comment_len] = 0;
    }
    sk->records++;
    sk->length += l;
    mo_sketch_add_seq(sk, seq, l);
}

/* kseq.h-style FASTA/FASTQ record reader over an in-memory (already inflated)
 * buffer: header = '>' or '@' line, name = up to first space/tab, comment = the
 * rest; sequence lines are concatenated until '>', '@' or '+'; after '+' the
 * quality is read until it is as long as the sequence.  Returns #records seen
 * (including short ones), or -1 on a truncated quality string. */
int64_t mo_sketch_add_fastx(mo_sketch *sk, const char *buf, size_t n)
{
    size_t p = 0; int64_t nrec = 0;
    char *seq = NULL; size_t seq_cap = 0;
    /* jump to first header */
    while (p < n && buf[p] != '>' && buf[p] != '@') p++;
    while (p < n) {
        /* header line */
        size_t h0 = p + 1, e = h0;
        while (e < n && buf[e] != '\n') e++;
        size_t hend = e; if (hend > h0 && buf[hend - 1] == '\r') hend--;
        size_t ne = h0; while (ne < hend && buf[ne] != ' ' && buf[ne] != '\t') ne++;
        size_t c0 = ne < hend ? ne + 1 : hend;
        p = e < n ? e + 1 : n;
        /* sequence lines */
        size_t l = 0;
        while (p < n && buf[p] != '>' && buf[p] != '+' && buf[p] != '@') {
            size_t le = p; while (le < n && buf[le] != '\n') le++;
            if (l + (le - p) + 1 > seq_cap) { seq_cap = (l + (le - p) + 1) * 2 + 256; seq = (char *)realloc(seq, seq_cap); }
            for (size_t q = p; q < le; q++) { unsigned char c = (unsigned char)buf[q]; if (c > ' ' && c != 127) seq[l++] = (char)c; }
            p = le < n ? le + 1 : n;
        }
        if (p < n && buf[p] == '+') {
            /* skip rest of '+' line, then quality until length l */
            while (p < n && buf[p] != '\n') p++;
            if (p < n) p++;
            size_t ql = 0;
            while (p < n && ql < l) {
                size_t le = p; while (le < n && buf[le] != '\n') le++;
                for (size_t q = p; q < le; q++) { unsigned char c = (unsigned char)buf[q]; if (c > ' ' && c != 127) ql++; }
                p = le < n ? le + 1 : n;
            }
            if (ql != l) { free(seq); return -1; }
            while (p < n && buf[p] != '>' && buf[p] != '@') p++;
        }
        nrec++;
        if (!seq) { seq_cap = 256; seq = (char *)malloc(seq_cap); }
        add_record(sk, buf + h0, ne - h0, buf + c0, hend - c0, seq, l);
    }
    free(seq);
    return nrec;
}

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

/* mash setMinHashesForReference / HashSet::toHashList + sort: ascending hashes
 * (and their counts).  Returns the number written (<= s). */
uint32_t mo_sketch_finish(const mo_sketch *sk, uint64_t *hashes, uint32_t *counts)
{
    uint32_t n = 0;
    for (uint64_t i = 0; i < sk->hashes.cap; i++) if (sk->hashes.cnt[i]) hashes[n++] = sk->hashes.keys[i];
    qsort(hashes, n, sizeof(uint64_t), cmp_u64);
    if (counts) for (uint32_t i = 0; i < n; i++) counts[i] = map_count(&sk->hashes, hashes[i]);
    return n;
}

/* mash MinHashHeap::estimateSetSize / estimateMultiplicity */
double mo_sketch_set_size(const mo_sketch *sk)
{
    if (sk->hashes.n == 0) return 0.0;
    return pow(2.0, sk->use64 ? 64.0 : 32.0) * (double)sk->hashes.n / (double)sk->queue.a[0];
}
double mo_sketch_multiplicity(const mo_sketch *sk)
{
    return sk->hashes.n ? (double)sk->multiplicity_sum / (double)sk->hashes.n : 0.0;
}
uint64_t mo_sketch_records(const mo_sketch *sk) { return sk->records; }
uint64_t mo_sketch_length(const mo_sketch *sk) { return sk->length; }
uint64_t mo_sketch_kmers(const mo_sketch *sk) { return sk->kmers_hashed; }
int mo_sketch_skipped_short(const mo_sketch *sk) { return sk->skipped_short; }
const char *mo_sketch_first_name(const mo_sketch *sk) { return sk->first_name; }
const char *mo_sketch_first_comment(const mo_sketch *sk) { return sk->first_comment; }

/* Independent brute-force definition of the same result, used to cross-check the
 * heap restatement: all window hashes of `seq` appended to `out` (caller sorts,
 * run-length-counts, keeps count >= m, truncates to s). Returns #hashes. */
uint64_t mo_all_window_hashes(const char *seq_in, uint64_t length, int k, uint64_t *out)
{
    if (length < (uint64_t)k) return 0;
    uint64_t n = 0;
    char fwd[64], rc[64];
    for (uint64_t i = 0; i + k <= length; i++) {
        int ok = 1;
        for (int t = 0; t < k; t++) {
            char c = seq_in[i + t];
            if (c > 96 && c < 123) c -= 32;
            char r;
            switch (c) { case 'A': r = 'T'; break; case 'C': r = 'G'; break;
                         case 'G': r = 'C'; break; case 'T': r = 'A'; break; default: ok = 0; r = 0; }
            fwd[t] = c; rc[k - 1 - t] = r;
        }
        if (!ok) continue;
        out[n++] = mo_kmer_hash(memcmp(fwd, rc, k) <= 0 ? fwd : rc, k, 42);
    }
    return n;
}

/* mash CommandDistance compareSketches(): two-pointer merge of ascending unique
 * lists, stop at sketchSize union elements, tail completion, Jaccard -> distance. */
void mo_compare(const uint64_t *ref, uint64_t nref, const uint64_t *qry, uint64_t nqry,
                uint64_t sketch_size, int k, uint64_t *common_out, uint64_t *denom_out, double *dist_out)
{
    uint64_t i = 0, j = 0, common = 0, denom = 0;
    while (denom < sketch_size && i < nref && j < nqry) {
        if (ref[i] < qry[j]) i++;
        else if (qry[j] < ref[i]) j++;
        else { i++; j++; common++; }
        denom++;
    }
    if (denom < sketch_size) {
        if (i < nref) denom += nref - i;
        if (j < nqry) denom += nqry - j;
        if (denom > sketch_size) denom = sketch_size;
    }
    double distance, jaccard = (double)common / (double)denom;
    if (common == denom) distance = 0.0;
    else if (common == 0) distance = 1.0;
    else {
        distance = -log(2.0 * jaccard / (1.0 + jaccard)) / (double)k;
        if (distance > 1.0) distance = 1.0;
    }
    *common_out = common; *denom_out = denom; *dist_out = distance;
}
