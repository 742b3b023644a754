// mhx_internal.h -- host-side internals of libmhx (not part of the C ABI).
#pragma once
#include <stdint.h>
#include <string>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/mhx.h"

namespace mhx {

// ---- errors ---------------------------------------------------------------------------
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// ---- sketch container (mhx_msh.cpp) -----------------------------------------------------
struct RefSketch {
    std::string name, comment;
    uint64_t length = 0;
    std::vector<uint64_t> hashes; // ascending; values < 2^32 when the sketch uses 32-bit hashes
    std::vector<uint32_t> counts; // optional (empty unless present in the file)
    // msh_parse_image(..., views = true): a 64-bit hash list stays where it is in the caller's file image (`hashes` empty)
    const uint64_t *view = nullptr;
    uint32_t view_n = 0;
    const uint64_t *hash_data() const { return view ? view : hashes.data(); }
    size_t hash_count() const { return view ? view_n : hashes.size(); }
};
struct SketchSet {
    uint32_t kmer_size = 0, sketch_size = 0, window_size = 0, hash_seed = 42;
    bool concatenated = true, noncanonical = false, preserve_case = false;
    float error = 0.f;
    std::string alphabet = "ACGT";
    std::vector<RefSketch> refs;
    bool use64() const { return kmer_size > 16; } // 4^k > 2^32
};
// batched distances with every row where it lies in host memory (mhx_engine.cpp; see dist_batch_core)
int dist_batch_rows(const uint64_t *const *q_rows, const uint32_t *q_len, uint32_t nq, const uint64_t *const *r_rows, const uint32_t *r_len,
                    uint32_t nr, int k, uint32_t s, uint32_t *common, uint32_t *denom, double *dist);
int msh_serialize(const SketchSet &s, std::vector<uint8_t> &out);
int msh_write_file(const char *path, const SketchSet &s);
int msh_read_file(const char *path, SketchSet &s);
// The container held in memory.  views: 64-bit hash lists are not copied out (RefSketch::view points into `raw`, which
// must be 8-byte aligned and outlive `s`), and their order is NOT checked here -- the caller does that (check_ascending),
// over the references in parallel if it likes.  `path` only names the file in messages.
int msh_parse_image(const uint8_t *raw, size_t n, const char *path, SketchSet &s, bool views);
bool check_ascending(const uint64_t *h, size_t n);

// ---- FASTA/FASTQ ingest (mhx_fastx.cpp) -------------------------------------------------
int read_all_maybe_gz(const char *path, std::vector<uint8_t> &out);
bool fastq_tail_complete(const uint8_t *t, size_t n); // is the last record of a 4-line FASTQ complete in kseq's sense? (mhx_fastx.cpp)
// Host threads the ingest may use (inflate, pread): MHX_INGEST_THREADS, else the cores this process may run on
// (sched_getaffinity, twice the cgroup's CPU quota if it has one) divided by the ranks of this node (LOCAL_WORLD_SIZE: one
// process per GPU), at most 32, at least 2.
int ingest_thread_budget();
struct ParsedRecords {
    std::vector<uint8_t> seq;  // bases of counted records, '\n' after each record (MHX_FMT_SEQ)
    uint64_t records = 0;      // records with length >= k (mash's `count`)
    uint64_t records_seen = 0; // all records
    uint64_t total_length = 0; // sum of counted record lengths
    bool skipped_short = false;
    std::string first_name, first_comment;
};
// kseq-compatible record reader over an inflated buffer (FASTA, FASTQ, multi-line either)
int parse_fastx(const uint8_t *buf, size_t n, int k, ParsedRecords &out);
bool looks_like_fastq4(const uint8_t *buf, size_t n);
void first_header(const uint8_t *buf, size_t n, std::string &name, std::string &comment);

// ---- gzip / DEFLATE decoder of the FASTQ ingest (mhx_inflate.cpp) ---------------------------
class GzInflater {
  public:
    static constexpr size_t kWindow = 32768;  // history a match may reach back into
    static constexpr size_t kOvershoot = 320; // inflate() may write this far past its limit
    GzInflater();
    ~GzInflater();
    GzInflater(const GzInflater &) = delete;
    GzInflater &operator=(const GzInflater &) = delete;
    static constexpr size_t kInputPad = 64;   // readable (zero) bytes the caller keeps behind the compressed input
    // the compressed file (all members); kInputPad readable bytes must follow data[n - 1]
    void set_input(const uint8_t *data, size_t n);
    void set_verify_crc(bool on);
    // Deferred checking: inflate() then returns at every member end (the member's last byte is the last byte it
    // produced) and take_member_end() hands out the CRC-32 the trailer promises, for a caller that computes the
    // CRC of the output itself (the ingest does so on a helper thread, behind the decoder).  The length is still
    // checked here.
    void set_deferred_crc(bool on);
    bool take_member_end(uint32_t *crc);
    // Produces output at `out` until `limit` bytes are reached (it may run over by < kOvershoot), the
    // input is exhausted or an error occurs; call again to continue.  [window_start, out) must hold the
    // previous output (up to 32 KiB of it).  Returns the bytes produced or (size_t)-1.
    size_t inflate(uint8_t *out, size_t limit, const uint8_t *window_start);
    bool done() const;
    const std::string &error() const;

  private:
    struct Impl;
    Impl *impl_;
};
uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n);

// ---- one gzip member decoded by several threads (mhx_pinflate.cpp) -----------------------------------------------
class ParallelGunzip {
  public:
    ParallelGunzip();
    ~ParallelGunzip();
    ParallelGunzip(const ParallelGunzip &) = delete;
    ParallelGunzip &operator=(const ParallelGunzip &) = delete;
    // The compressed file (GzInflater::kInputPad readable bytes behind data[n - 1]).  false: not worth it or not possible
    // (small input, no second block start found, not gzip): use the sequential GzInflater.
    // min_bytes / seg_bytes (0: defaults of 8 MiB / 1 MiB): smaller inputs are declined / target size of a segment.
    bool start(const uint8_t *data, size_t n, int threads, size_t min_bytes = 0, size_t seg_bytes = 0);
    // Output of the FIRST member, in order: bytes copied, 0 at its end (CRC-32 and length verified), (size_t)-1 on error.
    size_t read(uint8_t *dst, size_t want);
    // after the end: offset just behind the member's trailer (further members / padding follow there)
    size_t consumed_input() const;
    const std::string &error() const;

  private:
    struct Impl;
    Impl *impl_;
};

// ---- BGZF (bgzip): a gzip file of independent members of <= 64 KiB, each announcing its size (mhx_pinflate.cpp) ------
// mash reads such a file like any multi-member gzip (zlib's gzread); here the members are decoded side by side.
class BgzfReader {
  public:
    BgzfReader();
    ~BgzfReader();
    BgzfReader(const BgzfReader &) = delete;
    BgzfReader &operator=(const BgzfReader &) = delete;
    // The compressed file (GzInflater::kInputPad readable bytes behind data[n - 1]).  false: the file does not start
    // with a run of BGZF blocks worth the threads (use the other decoders).
    bool start(const uint8_t *data, size_t n, int threads);
    // Output of that run of blocks, in order: bytes copied, 0 at its end (every block's CRC-32 and length verified),
    // (size_t)-1 on error.
    size_t read(uint8_t *dst, size_t want);
    // after the end: offset just behind the last BGZF block (other members / padding may follow there)
    size_t consumed_input() const;
    const std::string &error() const;

  private:
    struct Impl;
    Impl *impl_;
};

// ---- statistics / text (mhx_text.cpp) ---------------------------------------------------
double binomial_cdf(uint64_t x, double p, uint64_t n);        // P[X <= x]
double binomial_sf_ge(uint64_t x, double p, uint64_t n);      // P[X >= x]
std::string fmt_g(double v);
std::string bounds_text(int k, double prob);

// Worker threads that are joined whatever happens: a std::thread destroyed while joinable ends the process in
// std::terminate, so neither an exception between creation and join nor a thread that could not be created (spawn
// returns false: the caller does that share of the work itself, or gives up) may leave one behind.
struct JoinedThreads {
    std::vector<std::thread> t;
    template <class... A> bool spawn(A &&...a)
    {
        try {
            t.emplace_back(std::forward<A>(a)...);
            return true;
        } catch (const std::system_error &) {
            return false;
        } catch (const std::bad_alloc &) {
            return false;
        }
    }
    void join()
    {
        for (auto &x : t)
            if (x.joinable()) x.join();
    }
    ~JoinedThreads() { join(); }
};

} // namespace mhx
