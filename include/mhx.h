/*
 * mhx.h -- C ABI of the MI355X-native MinHash sketch + distance engine ("mhx").
 *
 * This library is the in-process replacement for the `mash` child processes that
 * AuriClass spawns.  Every entry point names the reference interface it replaces
 * (paths relative to /root/reference):
 *
 *   mhx_sketch_files  <- `mash sketch [-r -m M] -o OUT -k K -s S files...`
 *                        auriclass/classes.py:576-596 (FASTQ) and :696-713 (FASTA)
 *   mhx_dist_files    <- `mash dist REF.msh QUERY.msh`      auriclass/classes.py:92-104
 *   mhx_bounds        <- `mash bounds -k K -p P`            auriclass/classes.py:305-318
 *   mhx_init          <- `mash -h` dependency probe         auriclass/general.py:198-205
 *
 * The remaining entry points expose the same hot path at buffer granularity (device
 * or host pointers) for the throughput benchmark, the multi-GPU shard/merge step and
 * the batched all-vs-refs distance (BASELINE.json configs 3-5).
 *
 * Conventions: plain C types only; the caller owns every buffer; functions return
 * MHX_OK (0) or a negative MHX_E_* code and leave a message for mhx_last_error().
 * Text outputs use the two-call pattern: pass cap = 0 to learn the size in *need
 * (including the terminating NUL), then call again with a buffer of that size.
 * All compute runs on the GPU selected by mhx_init(); there is no CPU fallback:
 * without a usable HIP device every compute entry point fails with MHX_E_NO_DEVICE.
 *
 * One device, one caller: the engine state (device, stream, staging buffers) is a process-wide
 * singleton, as `mash` was one process per sample (docs/running_analysis.md:47-59 of the reference).
 * mhx_init(d) binds the process to GPU d; a second mhx_init with another device tears the first
 * engine down.  Calls are not thread-safe against each other (only mhx_last_error is thread-local).
 * Several GPUs = several processes, one per GPU (auriclass_amd/multigpu.py; bench.py --gpus N).
 */
#ifndef MHX_H
#define MHX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHX_OK 0
#define MHX_E_NO_DEVICE (-1)   /* no HIP device / mhx_init not called                  */
#define MHX_E_ARG (-2)         /* bad argument                                          */
#define MHX_E_IO (-3)          /* file cannot be opened / read / written                */
#define MHX_E_NO_RECORDS (-4)  /* "ERROR: Did not find fasta records in ..." (mash)     */
#define MHX_E_FORMAT (-5)      /* malformed FASTA/FASTQ/.msh                            */
#define MHX_E_HIP (-6)         /* HIP runtime error                                     */
#define MHX_E_CAPACITY (-7)    /* caller buffer too small / device table exhausted      */
#define MHX_E_MISMATCH (-8)    /* sketches with different k / seed (mash refuses too)   */
#define MHX_E_INTERNAL (-9)

/* input stream formats for mhx_sketcher_push_* */
#define MHX_FMT_SEQ 0     /* dense sequence bytes; any non-ACGT byte (e.g. '\n') ends a k-mer run */
#define MHX_FMT_FASTQ4 1  /* strict 4-line FASTQ records, parsed on the device                      */

/* ---- library ---------------------------------------------------------------------- */
int mhx_init(int device);              /* selects the GPU, creates the stream; idempotent */
void mhx_shutdown(void);
const char *mhx_last_error(void);      /* thread-local message of the last failure        */
const char *mhx_version(void);
int mhx_device_name(char *buf, size_t cap);

/* ---- file level: what classes.py calls today through subprocess ---------------------- */

/* `mash sketch`.  reads != 0 => `-r -m min_mult`: all files form ONE reference whose
 * length is the set-size estimate; reads == 0 => one reference per file.  Writes the
 * unpacked Cap'n Proto sketch to out_msh and mash's stderr text (incl. the
 * "Estimated genome size: %g" line in reads mode) to stderr_buf.
 * est_genome_size may be NULL. */
int mhx_sketch_files(const char *const *paths, int n_paths, int k, uint32_t s, int reads,
                     uint32_t min_mult, const char *out_msh, char *stderr_buf, size_t stderr_cap,
                     size_t *stderr_need, double *est_genome_size);

/* `mash dist REF QUERY` stdout: rows "ref\tquery\tdist\tp\tcommon/denom\n", query-major. */
int mhx_dist_files(const char *ref_msh, const char *qry_msh, char *stdout_buf, size_t cap, size_t *need);

/* `mash bounds -k K -p P` stdout. */
int mhx_bounds(int k, double p, char *buf, size_t cap, size_t *need);

/* FASTA base count (replaces pyfastx.Fasta(f).size, classes.py:746-751). */
int mhx_fasta_total_bases(const char *path, uint64_t *total);
/* format sniffing (replaces pyfastx probes, general.py:68-115): 1 yes, 0 no, <0 error */
int mhx_sniff_fastq(const char *path);
int mhx_sniff_fasta(const char *path);
/* the end of a 4-line FASTQ stream (its last bytes, up to 64 KiB are enough): 1 when the last record is complete in kseq's
 * sense, 0 when it has its '+' line but no, or a differently long, quality string (kseq_read: -2; such a file is refused
 * by mhx_sketch_files); callers that push byte ranges of a file themselves ask here for its tail */
int mhx_fastq_tail_complete(const void *tail, size_t n);

/* ---- buffer level: the hot path itself ------------------------------------------------ */
typedef struct mhx_sketcher mhx_sketcher;

/* expected_bytes: upper bound of the bytes that will be pushed (sizes the device
 * candidate table and the initial admission threshold); 0 = unknown/small. */
int mhx_sketcher_create(int k, uint32_t s, uint32_t min_mult, uint64_t expected_bytes, mhx_sketcher **out);
/* Same with budget_scale times the candidate table and admission budget: what a caller asks for after
 * finish() has reported MHX_E_CAPACITY (inputs whose solid k-mers are fewer than s when min_mult > 1;
 * the file-level call retries with 16, 256, ... by itself). */
int mhx_sketcher_create_scaled(int k, uint32_t s, uint32_t min_mult, uint64_t expected_bytes, uint32_t budget_scale,
                               mhx_sketcher **out);
void mhx_sketcher_destroy(mhx_sketcher *sk);
int mhx_sketcher_reset(mhx_sketcher *sk);

/* Feed one record-aligned span.  The device pointer must be readable up to the next
 * 16-byte boundary past n (true for any hipMalloc/torch allocation).  Asynchronous on
 * the engine's stream; the buffer must stay valid AND unchanged until mhx_sketcher_finish() or
 * mhx_sketcher_sync() has returned -- a synchronisation of the stream alone is not enough: FASTQ spans
 * whose reads are longer than ~2.7 kb are read a second time by a pass that those two calls start. */
int mhx_sketcher_push_device(mhx_sketcher *sk, const void *d_bytes, uint64_t n, int fmt);
int mhx_sketcher_push_host(mhx_sketcher *sk, const void *h_bytes, uint64_t n, int fmt);
int mhx_sketcher_sync(mhx_sketcher *sk);

/* Final sketch: the s smallest distinct hashes with multiplicity >= min_mult, ascending.
 * hashes/counts must hold s entries (counts may be NULL). */
int mhx_sketcher_finish(mhx_sketcher *sk, uint64_t *hashes, uint32_t *counts, uint32_t *n_out);

/* stats of the pushes so far: [0] k-mers hashed, [1] table inserts, [2] lines seen (FASTQ4),
 * [3] device flags, [4] occupied table slots, [5] hash-kernel ms (profiling on), [6] launches,
 * [7] threshold */
int mhx_sketcher_stats(mhx_sketcher *sk, uint64_t *stats8);
/* FASTQ4 pushes so far: records whose sequence line holds >= k bytes -- the sequences `mash sketch`
 * counts (it skips shorter ones) and reports as "[N seqs]" in the .msh comment */
int mhx_sketcher_record_count(mhx_sketcher *sk, uint64_t *records);
int mhx_sketcher_debug_stamps(mhx_sketcher *sk, uint64_t *out8); /* per-phase cycle sums of a -DMHX_STAMPS diagnostic build */
int mhx_set_profiling(int on);   /* time hash-kernel launches with HIP events on the engine stream */
void *mhx_stream(void);          /* the engine's hipStream_t */

/* Multi-GPU: a shard's partial result = every (hash, count) it saw with hash <= limit
 * (no multiplicity filter).  export_threshold() returns the shard's own admission
 * threshold; ranks exchange the minimum, export with it, all-gather the slabs and merge. */
int mhx_sketcher_threshold(mhx_sketcher *sk, uint64_t *threshold);
int mhx_sketcher_export(mhx_sketcher *sk, uint64_t limit, uint64_t *hashes, uint32_t *counts,
                        uint32_t cap, uint32_t *n_out);
/* Same partial result, left on the device as one slab of int64 words ready for an all-gather:
 * [0] n (entries found; only min(n, cap) are stored), [1] the shard's threshold, [2] device flags,
 * [3, 3+cap) hashes, then cap/2 words holding the u32 counts; every entry <= the threshold, unsorted,
 * no multiplicity filter.  cap must be even.  The hash value 2^64-1 is not representable here: callers
 * fall back to mhx_sketcher_export when a threshold of 2^64-1 comes back. */
int mhx_sketcher_export_slab(mhx_sketcher *sk, void *d_slab, uint32_t cap);
/* The sharded path as the ranks run it (auriclass_amd/multigpu.py; SURVEY.md 8(e): sizes first, slabs sized from the
 * data, merge where the data is).  Replaces nothing in the reference by itself: it is how `mash sketch -r -m M` over ONE
 * sample (auriclass/classes.py:576-596) is spread over the GPUs of a node.
 *   1. mhx_sketcher_export_begin: every (hash, count) of this shard with hash <= its threshold T_r (no multiplicity
 *      filter) is compacted into a device buffer of the sketcher; header8 receives [0] n_r, [1] T_r, [2] device flags |
 *      MHX_SLAB_* bits, [3] occurrences of the hash value 2^64-1, [4] occupied table slots, [5..7] 0.  Ranks all-gather
 *      these 64 bytes.
 *   2. mhx_sketcher_export_pack: the entries as one slab of 8-byte words, hashes[cap_entries] then the u32 counts
 *      (cap_entries even, >= n_r; normally max_r n_r rounded up), written to dst -- device memory (RCCL send buffer) or
 *      host memory (gloo).  Entries beyond n_r are unspecified.  Ranks all-gather the slabs.
 *   3. mhx_sketcher_merge_slabs: slabs = the n_ranks gathered slabs back to back (slabs_on_device != 0: device
 *      memory), headers = the n_ranks gathered headers (8 words each).  The other ranks' entries <= T_min = min_r T_r
 *      are added to this rank's candidate table on the device and the union's sketch is extracted: the first s hashes
 *      with summed multiplicity >= min_mult.  MHX_E_CAPACITY if fewer than s qualify below a lowered T_min (every rank
 *      gets the same verdict from the same gathered data and sketches its shard again with a larger budget_scale).
 *      The sketcher's table now holds other shards' entries: mhx_sketcher_reset() before it is used again. */
int mhx_sketcher_export_begin(mhx_sketcher *sk, uint64_t *header8);
int mhx_sketcher_export_pack(mhx_sketcher *sk, void *dst, uint64_t cap_entries);
int mhx_sketcher_merge_slabs(mhx_sketcher *sk, const void *slabs, int slabs_on_device, uint32_t n_ranks,
                             uint64_t cap_entries, const uint64_t *headers, uint32_t own_rank, uint64_t *hashes,
                             uint32_t *counts, uint32_t *n_out);
/* The same exchange in ONE collective, for slabs that live on the device (RCCL): the header rides in front of the slab,
 * [header8 | hashes[cap_entries] | counts u32[cap_entries]] = 8 + cap_entries + cap_entries / 2 words.
 * mhx_sketcher_export_into compacts the shard's partial result straight into the caller's send buffer (header8 as for
 * export_begin; entries beyond cap_entries are counted in [0] but not stored).  After the all-gather,
 * mhx_sketcher_merge_gathered reads the headers back, and either merges (as mhx_sketcher_merge_slabs) or, when some shard
 * holds more entries than the slabs have room for, returns MHX_E_CAPACITY with *need_cap = the largest n_r: every rank
 * gets the same answer from the same gathered headers and repeats both calls with a larger cap_entries.  *need_cap == 0
 * with MHX_E_CAPACITY is the exactness rule's verdict (re-sketch with a larger budget_scale). */
int mhx_sketcher_export_into(mhx_sketcher *sk, void *d_slab, uint64_t cap_entries, uint64_t *header8);
int mhx_sketcher_merge_gathered(mhx_sketcher *sk, const void *d_slabs, uint32_t n_ranks, uint64_t cap_entries,
                                uint32_t own_rank, uint64_t *hashes, uint32_t *counts, uint32_t *n_out,
                                uint64_t *need_cap);
int mhx_merge_partials(const uint64_t *hashes, const uint32_t *counts, uint64_t n, uint32_t s,
                       uint32_t min_mult, uint64_t *out_hashes, uint32_t *out_counts, uint32_t *n_out);
/* bits a shard adds to word [2] of its slab besides the device flags (diagnostics of the m > 1 phase) */
#define MHX_SLAB_BOUNDED 0x100      /* a host-imposed cap has limited the shard's threshold at least once  */
#define MHX_SLAB_ESTABLISHED 0x200  /* the threshold has since been lowered from solid (count >= m) hashes */
/* The merge step of the sharded path together with its exactness rule: hashes/counts are the shards'
 * exports back to back (shard r contributes shard_n[r] entries, all <= shard_threshold[r]).  Entries above
 * T_min = min_r shard_threshold[r] are dropped (only below it is every shard's list complete), counts of equal
 * hashes are summed, count >= min_mult kept, first s returned.  If fewer than s qualify although some shard
 * has rejected hashes (T_min below the largest hash value of this k), the union cannot be decided from these
 * partials: MHX_E_CAPACITY -- every rank sketches its shard again with a larger budget_scale
 * (mhx_sketcher_create_scaled) and the exchange is repeated; a short sketch is never returned in that case.
 * SURVEY.md 8(e); the m = 3 default of /root/reference/auriclass/args.py:128-134 is what makes this matter. */
int mhx_merge_shard_partials(const uint64_t *hashes, const uint32_t *counts, const uint64_t *shard_n,
                             const uint64_t *shard_threshold, uint32_t n_shards, int k, uint32_t s,
                             uint32_t min_mult, uint64_t *out_hashes, uint32_t *out_counts, uint32_t *n_out);

/* Batched all-vs-refs `mash dist` arithmetic on the device: q and r are row-major
 * [n][stride] ascending unique hash lists with q_len/r_len valid entries each.
 * Outputs are [nq][nr] row-major (query-major like mash): common, denom, distance.
 * device_ptrs != 0 => q, r, q_len, r_len, common, denom, dist are device pointers. */
int mhx_dist_batch(const uint64_t *q, const uint32_t *q_len, uint32_t nq, const uint64_t *r,
                   const uint32_t *r_len, uint32_t nr, uint32_t stride, int k, uint32_t s,
                   uint32_t *common, uint32_t *denom, double *dist, int device_ptrs);
double mhx_last_dist_kernel_ms(void);
/* diagnostics of the last mhx_dist_batch / mhx_dist_files call: -1 = the generic pair kernel did all the work (tiny batch),
 * else the number of (query batch, reference slice) blocks the all-vs-refs fast path gave up to it (0 for uniform hashes) */
int mhx_last_dist_fallback_blocks(void);

/* scalar pieces of the dist row (host): mash pValue() */
double mhx_p_value(uint64_t common, uint64_t len_ref, uint64_t len_qry, int k, uint64_t denom);

/* .msh container access for callers that hold sketches in memory */
int mhx_msh_write(const char *path, int k, uint32_t s, uint32_t n_refs, const char *const *names,
                  const char *const *comments, const uint64_t *lengths, const uint64_t *const *hashes,
                  const uint32_t *n_hashes);

/* gunzip of an in-memory .gz (all members) with the decoder the FASTQ ingest uses in place of zlib
 * (kseq's gzread inside `mash sketch`, auriclass/classes.py:588).  out == NULL or cap too small:
 * *out_n still receives the inflated size (MHX_E_CAPACITY in the second case). */
int mhx_gunzip_buffer(const void *gz, size_t n, void *out, size_t cap, size_t *out_n);
/* The same with `threads` decoding threads on the first gzip member (block-boundary search, symbolic decoding of the
 * unknown 32 KiB windows, resolution; the result is byte-identical or the call fails); small inputs and threads < 2 take
 * the sequential decoder. */
int mhx_gunzip_buffer_mt(const void *gz, size_t n, void *out, size_t cap, size_t *out_n, int threads);

#ifdef __cplusplus
}
#endif
#endif /* MHX_H */
