// mhx_files.cpp -- the file-level half of libmhx: FASTQ / FASTA ingest (bulk path for uncompressed
// files, inflate threads for .gz, whole-file record parser as the last resort) and the two calls that
// replace AuriClass's `mash sketch` / `mash dist` subprocesses
// (/root/reference/auriclass/classes.py:576-596, 696-713, 92-104).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mhx_device.h"
#include "mhx_engine_internal.h"
#include "mhx_internal.h"

using namespace mhx;

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
                    if (rc) { mhx_sketcher_destroy(sk); return rc; } // the record parser's verdict is final (not the device parser's MHX_E_FORMAT, which sends the file HERE)
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

// Host buffers of the chunk pipeline.  Pinned (hipHostMalloc) when possible: the copy to the device then runs at the
// link's rate on the copy stream while the host thread goes on, instead of being staged through the runtime's bounce
// buffer.  Pinning 32 MiB costs milliseconds, so the engine keeps the buffers between calls (Engine::ingest_pinned);
// at most `limit` are in use per call, producers wait for one to come back.
struct HostBuf {
    uint8_t *p = nullptr;
    bool pinned = false;
};

class BufPool {
  public:
    BufPool(size_t bytes, size_t limit) : bytes_(bytes), limit_(limit) {}
    ~BufPool()
    {
        for (auto &b : spare_) {
            if (b.pinned && g.ready && g.ingest_pinned.size() < Engine::kIngestPinnedKeep) g.ingest_pinned.push_back(b.p);
            else if (b.pinned) hipHostFree(b.p);
            else free(b.p);
        }
    }
    HostBuf take()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (!spare_.empty()) { HostBuf b = spare_.back(); spare_.pop_back(); return b; }
            if (allocated_ < limit_ || abort_) break; // (after an abort nobody gives buffers back: let the producer run out)
            cv_.wait(lk);
        }
        ++allocated_;
        if (!g.ingest_pinned.empty()) { HostBuf b{(uint8_t *)g.ingest_pinned.back(), true}; g.ingest_pinned.pop_back(); return b; }
        lk.unlock();
        HostBuf b;
        static const bool no_pin = getenv("MHX_INGEST_PAGEABLE") != nullptr;
        (void)hipSetDevice(g.device); // producers are threads of their own
        if (!no_pin && hipHostMalloc((void **)&b.p, bytes_, hipHostMallocDefault) == hipSuccess) b.pinned = true;
        else { (void)hipGetLastError(); b.p = (uint8_t *)malloc(bytes_); b.pinned = false; }
        if (!b.p) throw std::bad_alloc();
        return b;
    }
    void give(HostBuf b)
    {
        if (!b.p) return;
        { std::lock_guard<std::mutex> lk(m_); spare_.push_back(b); }
        cv_.notify_one();
    }
    void abort() { { std::lock_guard<std::mutex> lk(m_); abort_ = true; } cv_.notify_all(); }

  private:
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<HostBuf> spare_;
    size_t bytes_, limit_, allocated_ = 0;
    bool abort_ = false;
};

// One record-aligned piece of an inflated FASTQ.  The buffer keeps GzInflater::kWindow bytes of room in
// front of the data: the previous 32 KiB of the stream, which DEFLATE matches may still refer to.
struct IngestChunk {
    HostBuf buf;           // kWindow + kIngestChunk + slack bytes, not zero-filled; goes back to `pool` with the chunk
    BufPool *pool = nullptr;
    size_t size = 0;
    int file = 0;
    bool first_of_file = false;
    IngestChunk() = default;
    IngestChunk(const IngestChunk &) = delete;
    IngestChunk &operator=(const IngestChunk &) = delete;
    IngestChunk(IngestChunk &&o) noexcept { *this = std::move(o); }
    IngestChunk &operator=(IngestChunk &&o) noexcept
    {
        if (this != &o) {
            release();
            buf = o.buf; pool = o.pool; size = o.size; file = o.file; first_of_file = o.first_of_file;
            o.buf = HostBuf(); o.pool = nullptr;
        }
        return *this;
    }
    ~IngestChunk() { release(); }
    void release() { if (pool && buf.p) pool->give(buf); buf = HostBuf(); }
    uint8_t *data() { return buf.p + GzInflater::kWindow; }
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
    // chunk buffers go round (BufPool): fresh 32 MiB allocations cost more in page faults than the inflate that fills them
    explicit ChunkQueue(size_t max_buffers) : pool_(IngestChunk::alloc_bytes(), max_buffers) {}
    void take_buffer(IngestChunk &c) { c.release(); c.buf = pool_.take(); c.pool = &pool_; }
    void producer_started() { std::lock_guard<std::mutex> lk(m_); ++live_; }
    void producer_done() { std::lock_guard<std::mutex> lk(m_); --live_; ready_.notify_all(); }
    void abort() { { std::lock_guard<std::mutex> lk(m_); abort_ = true; room_.notify_all(); } pool_.abort(); }
    bool aborted() { std::lock_guard<std::mutex> lk(m_); return abort_; }

  private:
    std::mutex m_;
    std::condition_variable ready_, room_;
    std::deque<IngestChunk> q_;
    BufPool pool_;
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

// A compressed input as the decoders want it: all of its bytes in memory with `pad` readable zero bytes behind them.
// The file is MAPPED (its pages are touched by the decoding threads, side by side, as they reach them) in front of one
// anonymous page that provides the padding; reading 240 MB into a zero-filled vector first cost a .fq.gz of 3 M reads
// 70 ms of serial start-up before the first block was decoded.  Falls back to reading into a plain buffer.
class CompressedFile {
  public:
    CompressedFile() = default;
    CompressedFile(const CompressedFile &) = delete;
    CompressedFile &operator=(const CompressedFile &) = delete;
    ~CompressedFile()
    {
        if (map_) munmap(map_, map_len_);
        free(buf_);
    }
    bool open(const char *path, size_t pad)
    {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || sb.st_size < 0) { ::close(fd); return false; }
        n_ = (size_t)sb.st_size;
        const size_t page = 4096, body = (n_ + page - 1) & ~(page - 1), tail = (pad + page - 1) & ~(page - 1);
        static const bool no_map = getenv("MHX_NO_MMAP") != nullptr;
        if (n_ > 0 && S_ISREG(sb.st_mode) && !no_map) {
            void *base = mmap(nullptr, body + tail, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); // zero pages ...
            if (base != MAP_FAILED) {
                if (mmap(base, n_, PROT_READ, MAP_PRIVATE | MAP_FIXED, fd, 0) != MAP_FAILED) { // ... with the file laid over the front
                    map_ = base; map_len_ = body + tail; data_ = (const uint8_t *)base;
                    ::close(fd);
                    return true;
                }
                munmap(base, body + tail);
            }
        }
        buf_ = (uint8_t *)malloc(n_ + pad + 1);
        if (!buf_) { ::close(fd); return false; }
        size_t got = 0;
        while (got < n_) {
            const ssize_t r = pread(fd, buf_ + got, n_ - got, (off_t)got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        ::close(fd);
        memset(buf_ + got, 0, n_ + pad - got);
        data_ = buf_;
        return got == n_;
    }
    const uint8_t *data() const { return data_; }
    size_t size() const { return n_; }

  private:
    void *map_ = nullptr;
    size_t map_len_ = 0, n_ = 0;
    uint8_t *buf_ = nullptr;
    const uint8_t *data_ = nullptr;
};

// CRC-32 of a gzip member computed BEHIND the decoder: the inflate thread hands over every piece of output as
// soon as it exists (and the trailer's expectation at a member end); this thread keeps the running CRC.
class CrcFollower {
  public:
    CrcFollower() : th_([this] { run(); }) {}
    ~CrcFollower()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void piece(const uint8_t *p, size_t n) { push({p, n, 0, false}); }
    void member_end(uint32_t expected) { push({nullptr, 0, expected, true}); }
    void drain() // everything handed over so far has been read: its buffers may go elsewhere now
    {
        std::unique_lock<std::mutex> lk(m_);
        idle_.wait(lk, [&] { return q_.empty() && !busy_; });
    }
    bool failed() { std::lock_guard<std::mutex> lk(m_); return failed_; }

  private:
    struct Item { const uint8_t *p; size_t n; uint32_t expected; bool end; };
    void push(Item it)
    {
        { std::lock_guard<std::mutex> lk(m_); q_.push_back(it); }
        cv_.notify_one();
    }
    void run()
    {
        uint32_t crc = 0;
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !q_.empty() || stop_; });
                if (q_.empty()) return;
                it = q_.front();
                q_.pop_front();
                busy_ = true;
            }
            bool bad = false;
            if (it.end) { bad = crc != it.expected; crc = 0; }
            else crc = crc32_update(crc, it.p, it.n);
            {
                std::lock_guard<std::mutex> lk(m_);
                busy_ = false;
                failed_ = failed_ || bad;
            }
            idle_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, idle_;
    std::deque<Item> q_;
    bool stop_ = false, busy_ = false, failed_ = false;
    std::thread th_; // last member: started when the rest is ready
};

// Producer of one input file: inflates it (own DEFLATE decoder on the whole compressed file in memory;
// MHX_ZLIB_INFLATE=1 selects zlib's gzread instead; an uncompressed file is simply read) and cuts the
// stream into record-aligned chunks.
void inflate_fastq(const char *path, int file, bool force_zlib, int decode_threads, ChunkQueue *q, FileIngestState *st)
{
    const bool gz = is_gzip_file(path);
    const bool own = gz && !force_zlib && !getenv("MHX_ZLIB_INFLATE");
    gzFile g = nullptr;
    FILE *plain = nullptr;
    CompressedFile zfile;
    GzInflater inf;
    std::unique_ptr<CrcFollower> crc_thread; // own decoder only: the CRC pass runs beside the decoding, not after it
    // a big single-member .gz (what sequencers write) is decoded by several threads (mhx_pinflate.cpp); what follows its
    // first member, small files and anything the parallel decoder declines go through the sequential decoder
    std::unique_ptr<ParallelGunzip> par;
    size_t par_base = 0; // where in the file the member that `par` decodes begins
    std::unique_ptr<BgzfReader> bgzf; // bgzip output: independent blocks of <= 64 KiB, decoded side by side
    if (own) {
        if (!zfile.open(path, GzInflater::kInputPad)) { st->error = std::string("ERROR: could not open ") + path + " for reading"; q->producer_done(); return; }
        const size_t zn = zfile.size();
        if (decode_threads >= 2) {
            bgzf.reset(new BgzfReader());
            if (!bgzf->start(zfile.data(), zn, decode_threads)) bgzf.reset();
        }
        if (!bgzf && decode_threads >= 2) {
            par.reset(new ParallelGunzip());
            if (!par->start(zfile.data(), zn, decode_threads)) par.reset();
        }
        inf.set_input(zfile.data(), zn);
        if (!getenv("MHX_INLINE_CRC")) { inf.set_deferred_crc(true); crc_thread.reset(new CrcFollower()); }
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
        q->take_buffer(c);
        uint8_t *d = c.data();
        if (!tail.empty()) memcpy(d + carry_len - tail.size(), tail.data(), tail.size());
        t_alloc += secs(t0, now());
        t0 = now();
        size_t n = carry_len;
        bool eof = false;
        while (n < kIngestChunk) {
            long got;
            if (bgzf) {
                const size_t r = bgzf->read(d + n, kIngestChunk - n);
                if (r == (size_t)-1) got = -1;
                else if (r == 0) { // end of the run of BGZF blocks (each one's CRC and length verified): the rest, if any, sequentially
                    const size_t off = bgzf->consumed_input(), zn = zfile.size();
                    bgzf.reset();
                    inf.set_input(zfile.data() + off, zn - off);
                    produced = 0;
                    if (off >= zn) { eof = true; break; }
                    continue;
                } else {
                    got = (long)r;
                    produced += r;
                }
            } else if (par) {
                const size_t r = par->read(d + n, kIngestChunk - n);
                if (r == (size_t)-1) got = -1;
                else if (r == 0) { // end of a member (CRC and length verified): a further large member (`cat a.gz b.gz`) gets the
                                   // threads again, anything else goes to the sequential decoder
                    const size_t off = par_base + par->consumed_input(), zn = zfile.size();
                    par.reset();
                    produced = 0;
                    if (off < zn) {
                        par.reset(new ParallelGunzip());
                        if (par->start(zfile.data() + off, zn - off, decode_threads)) { par_base = off; continue; }
                        par.reset();
                    }
                    inf.set_input(zfile.data() + off, zn - off);
                    continue;
                } else {
                    got = (long)r;
                    produced += r;
                }
            } else if (own) {
                const uint64_t hist = produced < GzInflater::kWindow ? produced : GzInflater::kWindow;
                // pieces of 1 MiB when the CRC follows on its own thread (it reads them while they are still in cache)
                const size_t want = crc_thread ? std::min<size_t>(kIngestChunk - n, 1u << 20) : kIngestChunk - n;
                const size_t r = inf.inflate(d + n, want, d + n - hist);
                if (r == (size_t)-1) got = -1;
                else {
                    got = (long)r;
                    produced += r;
                    if (crc_thread) {
                        uint32_t expected;
                        if (r) crc_thread->piece(d + n, r);
                        if (inf.take_member_end(&expected)) crc_thread->member_end(expected);
                    }
                    if (inf.done()) { n += r; eof = true; break; }
                    if (r == 0) continue; // a member ended exactly here: the next call starts the next one
                }
            } else if (gz) {
                got = gzread(g, d + n, (unsigned)std::min<size_t>(kIngestChunk - n, 1u << 30));
            } else {
                got = (long)fread(d + n, 1, kIngestChunk - n, plain);
                if (got == 0 && ferror(plain)) got = -1;
            }
            if (got < 0) {
                if (crc_thread) crc_thread->drain(); // the follower may still be reading pieces of this chunk's buffer
                st->error = std::string("ERROR: reading ") + path + " failed";
                st->own_inflate_failed = own; // the caller repeats the run with zlib before giving up
                close_all();
                q->producer_done();
                return;
            }
            if (got == 0) { eof = true; break; }
            n += (size_t)got;
        }
        if (crc_thread) {
            crc_thread->drain(); // the follower has read everything of this chunk: it may be cut, queued and recycled
            if (crc_thread->failed()) {
                st->error = std::string("ERROR: reading ") + path + " failed";
                st->own_inflate_failed = true;
                close_all();
                q->producer_done();
                return;
            }
        }
        t_inflate += secs(t0, now());
        t0 = now();
        if (first && n && d[0] != '@') { st->not_fastq4 = true; close_all(); q->producer_done(); return; }
        // Cut in front of the last record start that can be VERIFIED inside the chunk: a line that begins with '@', is
        // followed by a line that begins with neither '@' nor '+', and then by a line that begins with '+' -- the shape
        // no quality line can imitate (a quality line may begin with '@', but the line after it is a header and begins
        // with '@' as well).  A few lines are walked back from the end of the chunk; counting all of its newlines (round
        // 2) cost this thread a third of its time.  Every chunk thus starts at a record start; the records themselves
        // are parsed, checked for the 4-line layout and counted on the device.
        size_t cut = 0;
        if (!eof) {
            size_t ls[3] = {n, n, n}; // starts of the line under test and of the two lines behind it
            const uint8_t *p = (const uint8_t *)memrchr(d, '\n', n);
            for (int walked = 0; p && walked < 4096; ++walked) {
                const uint8_t *prev = p > d ? (const uint8_t *)memrchr(d, '\n', (size_t)(p - d)) : nullptr;
                ls[2] = ls[1]; ls[1] = ls[0];
                ls[0] = (size_t)(p - d) + 1;                 // the line that starts behind newline p ...
                const size_t start = prev ? (size_t)(prev - d) + 1 : 0; // ... and the one that ends with it
                // test the line [start, p]: its two successors start at ls[0] and ls[1]
                if (ls[1] < n && d[start] == '@' && d[ls[0]] != '@' && d[ls[0]] != '+' && d[ls[1]] == '+') { cut = start; break; }
                p = prev;
            }
            if (cut == 0) st->not_fastq4 = true; // no record start in sight (a record larger than a chunk, or not 4-line FASTQ): the record parser
        } else {
            // the last chunk starts at a record start like every other: its lines must come in fours (a last record may
            // lack its final newline)
            const uint64_t lines = count_newlines(d, n) + (n && d[n - 1] != '\n' ? 1 : 0);
            if (lines & 3) st->not_fastq4 = true;
            const size_t look = std::min<size_t>(n, 1u << 16);
            if (!fastq_tail_complete(d + n - look, look)) st->not_fastq4 = true; // quality string of the last record cut short: the record parser reports it
            cut = n;
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

// thread entry of a producer: nothing may escape it (a failed allocation ends the file's stream with an error, not the process)
void inflate_fastq_guarded(const char *path, int file, bool force_zlib, int decode_threads, ChunkQueue *q, FileIngestState *st)
{
    try {
        inflate_fastq(path, file, force_zlib, decode_threads, q, st);
    } catch (const std::exception &e) {
        st->error = std::string("ERROR: reading ") + path + " failed (" + e.what() + ")";
        q->abort();
        q->producer_done();
    }
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
    std::vector<uint8_t> tail; // its last bytes (is the last record complete?)
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
    JoinedThreads th;
    if (nthreads < 1) nthreads = 1;
    std::vector<int> ok((size_t)nthreads, 1);
    const size_t per = ((len + (size_t)nthreads - 1) / (size_t)nthreads + 4095) & ~(size_t)4095;
    for (int t = 0; t < nthreads; ++t) {
        const size_t b = (size_t)t * per;
        if (b >= len) break;
        const size_t e = std::min(len, b + per);
        const bool last = e == len; // the calling thread reads the last part itself (all of it when there is one part)
        auto part = [=, &ok]() {
            size_t done = b;
            while (done < e) {
                const ssize_t got = pread(fd, dst + done, e - done, (off_t)(off + done));
                if (got <= 0) { ok[(size_t)t] = 0; return; }
                done += (size_t)got;
            }
        };
        if (last || !th.spawn(part)) part(); // (or no thread to be had)
    }
    th.join();
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
    if (ensure_pinned_ring()) { clear_error(); (void)hipGetLastError(); return MHX_OK; } // no pinned staging: the chunked path will do
    int rc = MHX_OK;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(MHX_E_IO, "ERROR: could not open %s for reading", path);
    f->size = (uint64_t)sb.st_size;
    if (hipMalloc((void **)&f->d_buf, f->size + 64) != hipSuccess) { // tolerated: forget the sticky error, take the chunked path
        (void)hipGetLastError();
        close(fd);
        f->d_buf = nullptr;
        return MHX_OK;
    }
    int nthreads = std::min(16, ingest_thread_budget()); // pread threads: beyond 16 the page cache copy does not get faster
    uint64_t off = 0;
    int slot = 0;
    rc = MHX_OK;
    while (off < f->size && !rc) {
        const size_t len = (size_t)std::min<uint64_t>(kBulkBlock, f->size - off);
        if (hipEventSynchronize(g.pinned_free[slot]) != hipSuccess) { rc = fail(MHX_E_HIP, "pinned slot wait failed"); break; }
        if (!parallel_pread(fd, g.pinned[slot], off, len, len >= (8u << 20) ? nthreads : 1)) { rc = fail(MHX_E_IO, "ERROR: reading %s failed", path); break; }
        if (off == 0) f->head.assign(g.pinned[slot], g.pinned[slot] + std::min<size_t>(len, 1u << 20));
        if (off + len >= f->size) { // the last block: keep the file's last bytes (and what the block before it contributed, if this one is short)
            const size_t keep = std::min<size_t>(len, 1u << 16);
            f->tail.assign(g.pinned[slot] + len - keep, g.pinned[slot] + len);
        }
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
        if (!fastq_tail_complete(bf.tail.data(), bf.tail.size())) fallback = true; // a last record without its qualities: the record parser reports it
        if (!fallback) header.offer(i, bf.head.data(), bf.head.size(), k);
        if (!fallback) rc = mhx_sketcher_push_device(sk, bf.d_buf, bf.size, MHX_FMT_FASTQ4);
        if (hipStreamSynchronize(g.stream) != hipSuccess && !rc) rc = fail(MHX_E_HIP, "stream sync failed");
        bf.head.clear();
        resident.push_back(std::move(bf));
    }
    // 2. compressed files: one inflate thread per file (each with its share of decoding threads), 32 MiB record-aligned
    // chunks in pinned buffers; TWO device slots: chunk j is copied on the copy stream while chunk j - 1 is being parsed
    // and hashed on the engine stream.  A slot is overwritten only when the kernels of its previous chunk have completed
    // (event) AND that chunk is known to need no repair pass (sketcher_release_push) -- no whole-stream
    // synchronisation per chunk.
    std::vector<FileIngestState> st(n_paths);
    if (!rc && !fallback && !queued.empty()) {
        for (int i = 0; i < 2 && !rc; ++i) {
            if (!g.ingest_slot[i] && hipMalloc((void **)&g.ingest_slot[i], kIngestChunk + GzInflater::kOvershoot + 64) != hipSuccess) rc = fail(MHX_E_HIP, "hipMalloc failed for the ingest slot");
            if (!rc && !g.ingest_copied[i] && hipEventCreateWithFlags(&g.ingest_copied[i], hipEventDisableTiming) != hipSuccess) rc = fail(MHX_E_HIP, "event creation failed");
            if (!rc && !g.ingest_consumed[i] && hipEventCreateWithFlags(&g.ingest_consumed[i], hipEventDisableTiming) != hipSuccess) rc = fail(MHX_E_HIP, "event creation failed");
        }
        if (!rc && !g.ingest_word && hipHostMalloc((void **)&g.ingest_word, 64, hipHostMallocDefault) != hipSuccess) rc = fail(MHX_E_HIP, "hipHostMalloc failed");
        if (!rc && !g.copy_stream && hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(MHX_E_HIP, "stream creation failed");
    }
    if (!rc && !fallback && !queued.empty()) {
        ChunkQueue q(2 * queued.size() + 4);
        // (joined on every way out of this block; declared behind the queue they feed, so they are gone before it is)
        JoinedThreads threads;
        struct AbortOnExit { ChunkQueue &q; bool armed = true; ~AbortOnExit() { if (armed) { q.abort(); (void)hipStreamSynchronize(g.copy_stream); (void)hipStreamSynchronize(g.stream); } } } abort_guard{q}; // an exception in the loop below: the producers must not wait for room for ever
        for (size_t j = 0; j < queued.size(); ++j) q.producer_started();
        // decoding threads per .gz file: the host's share (ingest_thread_budget: MHX_INGEST_THREADS, else the cores this
        // process may run on divided by the ranks of the node) split over the files that are inflated side by side
        const int budget = ingest_thread_budget();
        const int per_file = std::max(1, budget / (int)std::max<size_t>(1, queued.size()) - 1);
        for (int i : queued) {
            if (threads.spawn(inflate_fastq_guarded, paths[i], i, force_zlib, per_file, &q, &st[i])) continue;
            q.producer_done(); // this file has no producer: give up on the call
            if (!rc) { rc = fail(MHX_E_INTERNAL, "could not start an ingest thread"); q.abort(); }
        }
        IngestChunk c, in_flight[2]; // in_flight[slot]: the host buffer whose copy into that slot may still be running
        uint64_t nchunk = 0;
        auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess && !rc) { rc = fail(MHX_E_HIP, "%s failed: %s", what, hipGetErrorString(e)); q.abort(); } return e == hipSuccess; };
        while (q.get(c)) {
            if (rc) continue; // drain
            if (c.first_of_file) header.offer(c.file, c.data(), std::min<size_t>(c.size, 1u << 20), k);
            const int slot = (int)(nchunk & 1);
            if (nchunk >= 2) { // the slot's previous tenant (chunk nchunk - 2): kernels through, repair question settled
                if (!hip_ok(hipEventSynchronize(g.ingest_consumed[slot]), "event wait")) continue;
                rc = sketcher_release_push(sk, g.ingest_slot[slot], g.copy_stream, g.ingest_word);
                if (rc) { q.abort(); continue; }
            }
            if (!hip_ok(hipMemcpyAsync(g.ingest_slot[slot], c.data(), c.size, hipMemcpyHostToDevice, g.copy_stream), "H2D copy")) continue;
            if (!hip_ok(hipEventRecord(g.ingest_copied[slot], g.copy_stream), "event record")) continue;
            if (!hip_ok(hipStreamWaitEvent(g.stream, g.ingest_copied[slot], 0), "stream wait")) continue;
            rc = mhx_sketcher_push_device(sk, g.ingest_slot[slot], c.size, MHX_FMT_FASTQ4);
            if (rc) { q.abort(); continue; }
            if (!hip_ok(hipEventRecord(g.ingest_consumed[slot], g.stream), "event record")) continue;
            // the host buffer goes back to the producers once its copy has completed: the one of the OTHER slot has had a
            // whole chunk's time for that
            if (in_flight[slot ^ 1].buf.p) {
                if (!hip_ok(hipEventSynchronize(g.ingest_copied[slot ^ 1]), "event wait")) continue;
                in_flight[slot ^ 1].release();
            }
            in_flight[slot] = std::move(c);
            ++nchunk;
        }
        if (hipStreamSynchronize(g.copy_stream) != hipSuccess && !rc) rc = fail(MHX_E_HIP, "copy stream sync failed");
        if (rc) hipStreamSynchronize(g.stream); // nothing may still read the slots or the buffers when we leave
        in_flight[0].release();
        in_flight[1].release();
        threads.join();
        abort_guard.armed = false;
    }
    bool own_failed = false;
    for (auto &f : st) own_failed = own_failed || f.own_inflate_failed;
    if (own_failed && !force_zlib) { // the engine's own decoder refused a stream: let zlib have the last word
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
    free_resident();
    if (sk) mhx_sketcher_destroy(sk);
    return rc;
}

// ---- FASTA through the device parser ------------------------------------------------------------------------
// `mash sketch` without -r: one reference per file (auriclass/classes.py:696-713).  The inflated file goes to the GPU
// as it is; mhx_fasta.hip squeezes it into the dense sequence stream (header lines dropped, line breaks removed inside a
// record, one separator byte in front of every record) and notes where the records start, the sketch kernel hashes the
// stream.  The host only looks at the record start positions (lengths, the count of records of >= k bases, the first
// of them for the reference's name).  Returns kFastaNotForDevice when the file is not plain FASTA (does not start
// with '>', or holds FASTQ syntax): the caller then takes the host record parser, as before.
namespace {
constexpr int kFastaNotForDevice = 1;

struct FastaInfo {
    uint64_t records = 0, total_length = 0;
    std::string first_name, first_comment;
};

struct DeviceBuf {
    void *p = nullptr;
    ~DeviceBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n); }
};

// header line number `idx` (0-based, counting lines that start with '>') of a FASTA held in memory
bool nth_header(const uint8_t *b, size_t n, uint64_t idx, std::string &name, std::string &comment)
{
    size_t p = 0;
    uint64_t seen = 0;
    while (p < n) {
        const uint8_t *e = (const uint8_t *)memchr(b + p, '\n', n - p);
        const size_t le = e ? (size_t)(e - b) : n;
        if (b[p] == '>') {
            if (seen == idx) { first_header(b + p, le - p, name, comment); return true; }
            ++seen;
        }
        p = le + 1;
    }
    return false;
}
} // namespace

// One FASTA file, bytes in host memory (a pinned staging slot or a vector) -> its reference sketch.  Device buffers and the
// sketcher live in g.fasta between files and calls; two stream synchronisations per file (record layout, sketch).
// device buffers of the FASTA path for files of up to n bytes (both raw buffers, the stream, the workspace), the pinned
// word block, the record positions, the two "raw bytes are in place" events
static int ensure_fasta_buffers(uint64_t n)
{
    FastaCtx &c = g.fasta;
    size_t os, oi, oo, of;
    const size_t ws_bytes = fasta_workspace_bytes(n, &os, &oi, &oo, &of);
    if (!c.h_words) HIPCHK(hipHostMalloc((void **)&c.h_words, (2 + (size_t)kFastaSepsInline) * sizeof(uint64_t), hipHostMallocDefault));
    for (int i = 0; i < 2; ++i)
        if (!c.raw_ready[i]) HIPCHK(hipEventCreateWithFlags(&c.raw_ready[i], hipEventDisableTiming));
    if (c.raw_cap < n + 64 || c.ws_cap < ws_bytes) { // grow with room: the next assembly is about as large as this one
        HIPCHK(hipStreamSynchronize(g.stream));
        if (g.copy_stream) HIPCHK(hipStreamSynchronize(g.copy_stream));
        hipFree(c.d_raw[0]); hipFree(c.d_raw[1]); hipFree(c.d_out); hipFree(c.d_ws);
        c.d_raw[0] = c.d_raw[1] = c.d_out = c.d_ws = nullptr;
        c.raw_cap = c.ws_cap = 0;
        const size_t want = (size_t)(n + n / 4 + (1u << 20));
        size_t os2, oi2, oo2, of2;
        const size_t ws_want = fasta_workspace_bytes(want, &os2, &oi2, &oo2, &of2);
        if (hipMalloc((void **)&c.d_raw[0], want + 64) != hipSuccess || hipMalloc((void **)&c.d_raw[1], want + 64) != hipSuccess ||
            hipMalloc((void **)&c.d_out, want + 64) != hipSuccess || hipMalloc((void **)&c.d_ws, ws_want) != hipSuccess)
            return fail(MHX_E_HIP, "hipMalloc failed for the FASTA buffers (%llu bytes)", (unsigned long long)n);
        c.raw_cap = want + 64;
        c.ws_cap = ws_want;
    }
    if (!c.d_seps) {
        c.seps_cap = 1u << 16;
        if (hipMalloc((void **)&c.d_seps, (size_t)c.seps_cap * 8) != hipSuccess) { c.seps_cap = 0; return fail(MHX_E_HIP, "hipMalloc failed for the FASTA record positions"); }
    }
    return MHX_OK;
}

// One FASTA file, bytes in host memory (a pinned staging slot or a vector) -> its reference sketch.  Device buffers and the
// sketcher live in g.fasta between files and calls; two stream synchronisations per file (record layout, sketch).
// which: the raw buffer of this file (d_raw[which]); on_device: the loader has already copied the bytes there on the copy
// stream (raw_ready[which] says when)
static int sketch_fasta_on_device(const uint8_t *raw, uint64_t n, int which, bool on_device, int k, uint32_t s, std::vector<uint64_t> &hashes, FastaInfo &info)
{
    if (n == 0 || raw[0] != '>' || n > 0x7FFFFFFF00ull) return kFastaNotForDevice;
    FastaCtx &c = g.fasta;
    int rc0 = ensure_fasta_buffers(n);
    if (rc0) return rc0;
    size_t os, oi, oo, of;
    fasta_workspace_bytes(n, &os, &oi, &oo, &of);
    const uint64_t ntiles = (n + kFastaTile - 1) / kFastaTile;
    uint8_t *d_raw = c.d_raw[which & 1];
    if (on_device) HIPCHK(hipStreamWaitEvent(g.stream, c.raw_ready[which & 1], 0));
    else HIPCHK(hipMemcpyAsync(d_raw, raw, n, hipMemcpyHostToDevice, g.stream));
    uint64_t total = 0;
    uint32_t nsep = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIPCHK(launch_fasta_compact(d_raw, n, c.d_ws, c.d_out, c.d_seps, c.seps_cap, g.stream));
        HIPCHK(hipMemcpyAsync(&c.h_words[0], c.d_ws + oo + 8 * ntiles, 8, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipMemcpyAsync(&c.h_words[1], c.d_ws + of, 8, hipMemcpyDeviceToHost, g.stream));
        // the record positions of an assembly (tens to hundreds of contigs) ride along with the same synchronisation
        HIPCHK(hipMemcpyAsync(&c.h_words[2], c.d_seps, (size_t)std::min(c.seps_cap, kFastaSepsInline) * 8, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        total = c.h_words[0];
        uint32_t fl[2];
        memcpy(fl, &c.h_words[1], 8);
        if (fl[0] & 1u) return kFastaNotForDevice;
        nsep = fl[1];
        if (nsep <= c.seps_cap) break;
        if (attempt) return fail(MHX_E_INTERNAL, "FASTA record list kept growing");
        hipFree(c.d_seps); // more records than there was room for: once more with room for all of them
        c.d_seps = nullptr;
        c.seps_cap = 0;
        const uint32_t want = nsep + nsep / 4 + 16;
        if (hipMalloc((void **)&c.d_seps, (size_t)want * 8) != hipSuccess) return fail(MHX_E_HIP, "hipMalloc failed for %u FASTA record positions", want);
        c.seps_cap = want;
    }
    // records: separator i sits in front of record i; its length is the distance to the next separator (or the end)
    std::vector<uint64_t> seps(nsep);
    if (nsep && nsep <= kFastaSepsInline) memcpy(seps.data(), &c.h_words[2], (size_t)nsep * 8);
    else if (nsep) {
        HIPCHK(hipMemcpyAsync(seps.data(), c.d_seps, seps.size() * 8, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    std::sort(seps.begin(), seps.end());
    info = FastaInfo();
    uint64_t first_counted = ~0ull;
    for (size_t i = 0; i < seps.size(); ++i) {
        const uint64_t len = (i + 1 < seps.size() ? seps[i + 1] : total) - seps[i] - 1;
        if (len >= (uint64_t)k) {
            if (first_counted == ~0ull) first_counted = i;
            ++info.records;
            info.total_length += len;
        }
    }
    // a last header line without its newline has no separator and no sequence: it does not count either way
    if (info.records == 0) return MHX_OK; // the caller reports "Did not find fasta records"
    nth_header(raw, n, first_counted, info.first_name, info.first_comment);
    uint64_t boost = 1;
    for (int attempt = 0; attempt < 6; ++attempt) {
        int rc = MHX_OK;
        if (!c.sk || c.k != k || c.s != s || c.scale != boost) { // one sketcher per (k, s), reset between files
            if (c.sk) mhx_sketcher_destroy(c.sk);
            c.sk = nullptr;
            rc = create_sketcher(k, s, 1, 0, boost, &c.sk);
            if (rc) return rc;
            c.k = k; c.s = s; c.scale = boost;
        } else {
            rc = mhx_sketcher_reset(c.sk);
            if (rc) return rc;
        }
        rc = mhx_sketcher_push_device(c.sk, c.d_out, total, MHX_FMT_SEQ);
        uint32_t nh = 0;
        if (!rc) {
            hashes.resize(s);
            rc = mhx_sketcher_finish(c.sk, hashes.data(), nullptr, &nh);
        }
        if (rc == MHX_E_CAPACITY) { boost *= 16; continue; }
        if (rc) return rc;
        hashes.resize(nh);
        return MHX_OK;
    }
    return fail(MHX_E_CAPACITY, "could not size the device table for this input");
}

// One input of `mash sketch` without -r, in host memory: an uncompressed file of up to 64 MiB is pread() straight into
// a pinned staging slot (no zero-fill, no pageable bounce on the way to the GPU), anything else is read / inflated into
// a vector.  Runs on a helper thread for file i + 1 while file i is on the GPU.
struct FastaInput {
    std::vector<uint8_t> owned;
    const uint8_t *data = nullptr;
    uint64_t n = 0;
    int slot = -1;
    bool on_device = false; // the bytes are already on their way into the file's raw buffer (raw_ready event)
    int rc = MHX_OK;
    std::string error; // (mhx_last_error is thread-local: the message travels with the input)
};

// d_dst != nullptr: the file's raw device buffer is free and large enough -- every reader thread sends what it has read
// on the copy stream right away (MiB by MiB), so the copy runs beside the reading instead of behind it
static void load_fasta_input_impl(const char *path, int slot, bool pinned_ok, FastaInput *in, uint8_t *d_dst, size_t d_cap, hipEvent_t ready);
static void load_fasta_input(const char *path, int slot, bool pinned_ok, FastaInput *in, uint8_t *d_dst, size_t d_cap, hipEvent_t ready)
{ // runs on a helper thread: nothing may escape it
    try {
        load_fasta_input_impl(path, slot, pinned_ok, in, d_dst, d_cap, ready);
    } catch (const std::exception &e) {
        in->rc = MHX_E_INTERNAL;
        in->error = std::string("reading ") + path + ": " + e.what();
    }
}

static void load_fasta_input_impl(const char *path, int slot, bool pinned_ok, FastaInput *in, uint8_t *d_dst, size_t d_cap, hipEvent_t ready)
{
    (void)hipSetDevice(g.device);
    struct stat sb;
    if (pinned_ok && !is_gzip_file(path) && stat(path, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && (size_t)sb.st_size + 64 <= kBulkBlock) {
        const int fd = open(path, O_RDONLY);
        if (fd >= 0) {
            bool ok = hipEventSynchronize(g.pinned_free[slot]) == hipSuccess;
            const size_t len = (size_t)sb.st_size;
            const bool stream = d_dst && len + 64 <= d_cap;
            const int nthreads = len >= (4u << 20) ? std::min(8, ingest_thread_budget()) : 1;
            uint8_t *dst = g.pinned[slot];
            JoinedThreads th;
            std::vector<int> good((size_t)nthreads, 1);
            const size_t per = ((len + (size_t)nthreads - 1) / (size_t)nthreads + 4095) & ~(size_t)4095;
            for (int t = 0; ok && t < nthreads; ++t) {
                const size_t b = (size_t)t * per;
                if (b >= len) break;
                const size_t e = std::min(len, b + per);
                auto part = [=, &good]() {
                    (void)hipSetDevice(g.device);
                    size_t done = b, sent = b;
                    while (done < e) {
                        const ssize_t got = pread(fd, dst + done, std::min<size_t>(e - done, 1u << 20), (off_t)done);
                        if (got <= 0) { good[(size_t)t] = 0; return; }
                        done += (size_t)got;
                        if (stream && (done - sent >= (1u << 20) || done == e)) {
                            if (hipMemcpyAsync(d_dst + sent, dst + sent, done - sent, hipMemcpyHostToDevice, g.copy_stream) != hipSuccess) { good[(size_t)t] = 0; return; }
                            sent = done;
                        }
                    }
                };
                if (!th.spawn(part)) part(); // no thread to be had: this one reads the part itself
            }
            th.join();
            for (int v : good) ok = ok && v;
            close(fd);
            if (ok && stream) ok = hipEventRecord(ready, g.copy_stream) == hipSuccess;
            if (ok) { in->data = dst; in->n = len; in->slot = slot; in->on_device = stream; return; }
            if (stream) (void)hipStreamSynchronize(g.copy_stream); // nothing of a failed load may still be in flight
            (void)hipGetLastError();
        }
    }
    in->rc = read_all_maybe_gz(path, in->owned);
    if (in->rc) in->error = mhx_last_error();
    in->data = in->owned.data();
    in->n = in->owned.size();
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
        // name / comment / count: the first counted record and the number of counted ones (sequence of at least
        // k bytes) -- from the record parser where it ran, from a 4-line walk over the raw bytes where the stream
        // went to the device parser untouched
        bool any = streamed;
        std::string fb_name, fb_comment;
        bool have_fb = false;
        for (auto &l : loaded) {
            if (streamed) break;
            if (l.rec.records_seen || !l.rec.seq.empty()) {
                if (!any && l.rec.records) { fname = l.rec.first_name; fcomment = l.rec.first_comment; any = true; }
                count += l.rec.records;
            } else if (!l.raw.empty()) {
                const uint8_t *b = l.raw.data();
                const size_t n = l.raw.size();
                if (!have_fb) { first_header(b, n, fb_name, fb_comment); have_fb = true; }
                size_t p = 0;
                while (p < n) { // one record: header, sequence, '+', quality
                    const uint8_t *e0 = (const uint8_t *)memchr(b + p, '\n', n - p);
                    if (!e0) break;
                    const size_t s0 = (size_t)(e0 - b) + 1;
                    const uint8_t *e1 = s0 < n ? (const uint8_t *)memchr(b + s0, '\n', n - s0) : nullptr;
                    const size_t s1 = e1 ? (size_t)(e1 - b) : n;
                    const size_t len = s1 - s0 - ((s1 > s0 && b[s1 - 1] == '\r') ? 1 : 0);
                    if (len >= (size_t)k) {
                        if (!any) { first_header(b + p, s1 - p, fname, fcomment); any = true; }
                        ++count;
                    }
                    size_t q = s1 + 1;
                    for (int i = 0; i < 2 && q < n; ++i) {
                        const uint8_t *e = (const uint8_t *)memchr(b + q, '\n', n - q);
                        q = e ? (size_t)(e - b) + 1 : n;
                    }
                    p = q;
                }
            }
        }
        if (!any && have_fb) { fname = fb_name; fcomment = fb_comment; }
        // mash stops when no record holds k bases.  (`kmers` is not that question: the device counts every window of k BYTES
        // inside one line, and a CRLF file of reads one base shorter than k has such windows -- 31 bases and the '\r' --
        // although none of them is a k-mer.)
        if (count == 0) return no_records(paths[0]);
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
        // one reference per file; file i + 1 is read (and inflated) on a helper thread while file i is on the GPU
        const bool pinned_ok = !getenv("MHX_HOST_FASTA") && ensure_pinned_ring() == MHX_OK;
        if (!pinned_ok) { clear_error(); (void)hipGetLastError(); }
        std::vector<FastaInput> inputs(n_paths);
        // the device buffers are sized for the largest uncompressed input up front, so that a loader can copy file i + 1
        // into its raw buffer while file i is still on the GPU
        uint64_t largest = 0;
        for (int i = 0; i < n_paths; ++i) {
            struct stat sb;
            if (stat(paths[i], &sb) == 0 && S_ISREG(sb.st_mode) && !is_gzip_file(paths[i])) largest = std::max<uint64_t>(largest, (uint64_t)sb.st_size);
        }
        bool stream_ok = pinned_ok && largest > 0 && largest + 64 <= kBulkBlock && ensure_fasta_buffers(largest) == MHX_OK;
        if (!stream_ok) clear_error();
        auto raw_of = [&](int i) { return stream_ok ? g.fasta.d_raw[i & 1] : (uint8_t *)nullptr; };
        std::thread loader;
        struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{loader};
        auto start_loader = [&](int j) { // file j on its way while file j - 1 is on the GPU; without a thread to be had: loaded here and now
            try {
                loader = std::thread(load_fasta_input, paths[j], j % Engine::kPinnedSlots, pinned_ok, &inputs[j], raw_of(j), g.fasta.raw_cap, g.fasta.raw_ready[j & 1]);
            } catch (const std::system_error &) {
                load_fasta_input(paths[j], j % Engine::kPinnedSlots, pinned_ok, &inputs[j], raw_of(j), g.fasta.raw_cap, g.fasta.raw_ready[j & 1]);
            }
        };
        load_fasta_input(paths[0], 0, pinned_ok, &inputs[0], raw_of(0), g.fasta.raw_cap, g.fasta.raw_ready[0]);
        for (int i = 0; i < n_paths; ++i) {
            if (loader.joinable()) loader.join();
            if (i + 1 < n_paths) start_loader(i + 1);
            FastaInput &in = inputs[i];
            err += std::string("Sketching ") + paths[i] + "...\n";
            if (in.rc) return fail(in.rc, "%s", in.error.c_str());
            if (!getenv("MHX_HOST_FASTA")) { // plain FASTA: parsed on the device
                RefSketch ref;
                FastaInfo info;
                rc = sketch_fasta_on_device(in.data, in.n, i, in.on_device, k, s, ref.hashes, info);
                if (rc < 0) return rc;
                if (rc == MHX_OK) {
                    if (info.records == 0) return no_records(paths[i]);
                    ref.name = paths[i];
                    ref.comment = make_comment(info.first_name, info.first_comment, info.records);
                    ref.length = info.total_length;
                    set.refs.push_back(std::move(ref));
                    in = FastaInput();
                    continue;
                }
            }
            if (in.slot >= 0) in.owned.assign(in.data, in.data + in.n); // the record parser path keeps the bytes beyond this slot's turn
            loaded[i].raw.swap(in.owned);
            in = FastaInput();
            rc = parse_fastx(loaded[i].raw.data(), loaded[i].raw.size(), k, loaded[i].rec);
            if (rc) return rc;
            if (loaded[i].rec.records == 0) return no_records(paths[i]);
            RefSketch ref;
            std::vector<Loaded *> in2{&loaded[i]};
            rc = sketch_reference(in2, k, s, 1, false, ref.hashes, ref.counts, nullptr);
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
    // The reference sketch file (9.6 MB at AuriClass's defaults: 24 x 50 000 hashes) is read ONCE, into a pinned block,
    // parsed where it is (64-bit hash lists stay views into the image), checked for order on a few threads and copied
    // row by row from the pinned image into the device staging area: one pass over the bytes on the host instead of
    // five (file buffer, segment copies, hash vectors, padded matrix, pageable H2D staging): 5.6 -> 2 ms per call.
    SketchSet R, Q;
    std::vector<uint8_t> ref_heap;
    static const bool timing = getenv("MHX_DIST_TIMING") != nullptr; // phase times of a call on stderr
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[mhx dist_files] %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    };
    {
        struct stat sb;
        const int fd = open(ref_msh, O_RDONLY);
        if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { if (fd >= 0) close(fd); return fail(MHX_E_IO, "cannot open sketch %s", ref_msh); }
        const size_t len = (size_t)sb.st_size;
        uint8_t *img = nullptr;
        if (len >= (1u << 20) && len <= (256u << 20)) { // a pinned block of its own, kept between calls (larger files, or MHX_DIST_PAGEABLE=1: the heap)
            if (g.dist_img_cap < len && !getenv("MHX_DIST_PAGEABLE")) {
                if (g.dist_img) hipHostFree(g.dist_img);
                g.dist_img = nullptr;
                g.dist_img_cap = 0;
                const size_t cap = (len + len / 4 + (1u << 20)) & ~(size_t)((1u << 20) - 1);
                if (hipHostMalloc((void **)&g.dist_img, cap, hipHostMallocDefault) == hipSuccess) g.dist_img_cap = cap;
                else { g.dist_img = nullptr; (void)hipGetLastError(); }
            }
            if (g.dist_img_cap >= len) img = g.dist_img;
        }
        if (!img) { ref_heap.resize(len); img = ref_heap.data(); }
        const bool ok = len == 0 || parallel_pread(fd, img, 0, len, len >= (4u << 20) ? std::min(8, ingest_thread_budget()) : 1);
        close(fd);
        if (!ok) return fail(MHX_E_IO, "cannot read %s", ref_msh);
        lap("reference file read");
        rc = msh_parse_image(img, len, ref_msh, R, true);
        if (rc) return rc;
        lap("parsed");
        // the distance kernels merge ascending duplicate-free lists (what mash writes); anything else is a damaged file
        std::vector<int> bad(R.refs.size(), 0);
        {
            JoinedThreads th;
            const size_t nthreads = len >= (4u << 20) ? (size_t)std::min(8, ingest_thread_budget()) : 1;
            for (size_t t = 0; t < nthreads; ++t) {
                auto part = [&, t]() {
                    for (size_t i = t; i < R.refs.size(); i += nthreads)
                        if (R.refs[i].view && !check_ascending(R.refs[i].view, R.refs[i].view_n)) bad[i] = 1;
                };
                if (t + 1 == nthreads || !th.spawn(part)) part();
            }
        }
        for (size_t i = 0; i < bad.size(); ++i)
            if (bad[i]) return fail(MHX_E_FORMAT, "%s: hash list of reference %zu is not ascending", ref_msh, i);
        lap("order checked");
    }
    rc = msh_read_file(qry_msh, Q);
    if (rc) return rc;
    lap("query read");
    if (R.kmer_size != Q.kmer_size)
        return fail(MHX_E_MISMATCH, "ERROR: The query and reference sketches have different k-mer sizes (%u and %u)", Q.kmer_size, R.kmer_size);
    if (R.hash_seed != Q.hash_seed) return fail(MHX_E_MISMATCH, "ERROR: The query and reference sketches have different hash seeds");
    const int k = (int)R.kmer_size;
    const uint32_t s = R.sketch_size < Q.sketch_size ? R.sketch_size : Q.sketch_size;
    const uint32_t nr = (uint32_t)R.refs.size(), nq = (uint32_t)Q.refs.size();
    std::string text;
    if (nr && nq) {
        std::vector<const uint64_t *> rrows(nr), qrows(nq);
        std::vector<uint32_t> rl(nr), ql(nq);
        for (uint32_t i = 0; i < nr; ++i) { rrows[i] = R.refs[i].hash_data(); rl[i] = (uint32_t)R.refs[i].hash_count(); }
        for (uint32_t i = 0; i < nq; ++i) { qrows[i] = Q.refs[i].hash_data(); ql[i] = (uint32_t)Q.refs[i].hash_count(); }
        std::vector<uint32_t> common((size_t)nq * nr), denom((size_t)nq * nr);
        std::vector<double> dist((size_t)nq * nr);
        rc = dist_batch_rows(qrows.data(), ql.data(), nq, rrows.data(), rl.data(), nr, k, s, common.data(), denom.data(), dist.data());
        if (rc) return rc;
        lap("distances back");
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
    try {
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
    } catch (const std::bad_alloc &) {
        return fail(MHX_E_INTERNAL, "mhx_msh_write: out of host memory");
    } catch (const std::exception &e) {
        return fail(MHX_E_INTERNAL, "mhx_msh_write: %s", e.what());
    }
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

