// mhx_fasta.hip -- FASTA on the device: the raw (inflated) file bytes go to HBM as they are and three small
// kernels turn them into the dense sequence stream (MHX_FMT_SEQ) the sketch kernel hashes:
//
//   fasta_scan_kernel     per 16 KiB tile: which bytes would be kept, for either state the tile may start in
//   fasta_offsets_kernel  one workgroup: resolves every tile's start state and output offset (scan over tiles)
//   fasta_compact_kernel  per tile: writes the kept bytes to their place, notes where every record starts
//
// What is kept is exactly what kseq + mash keep (mash `sketch` without -r, Sketch.cpp sketchFile; AuriClass hands the
// FASTA paths over untouched, /root/reference/auriclass/classes.py:696-713): a line that starts with '>' is a header,
// everything up to the next header is the record's sequence with every byte <= ' ' (line breaks, CR, blanks) and DEL
// squeezed out, so k-mers span the line breaks of a record; the header's own newline stays as the one separator byte in
// front of each record, so no k-mer spans two records.  A line that starts with '@' or '+' (FASTQ syntax) raises a flag and
// the host falls back to its record parser, as it does when the file does not start with '>'.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mhx_device.h"
#include "mhx_tile.h"

namespace mhx {

namespace {

constexpr int kFaThreads = 256;
constexpr int kFaBytesPerThread = 64;
constexpr int kFaTile = kFaThreads * kFaBytesPerThread; // 16 KiB

// state of the line that is open at some point of the stream
enum : uint32_t { kPass = 0, kSeq = 1, kHdr = 2 };
__device__ __forceinline__ uint32_t combine(uint32_t a, uint32_t b) { return b != kPass ? b : a; }

struct ChunkMasks {
    uint64_t in;    // bytes inside the file
    uint64_t nl;    // newline bytes
    uint64_t keepc; // bytes a sequence line keeps (> ' ' and not DEL)
    uint64_t ls;    // line starts: the previous byte is a newline (or the file begins here)
    uint64_t gt;    // '>' bytes
    uint64_t fq;    // '@' or '+' bytes
};

__device__ __forceinline__ uint64_t byte_eq_mask64(const uint32_t (&d)[16], uint32_t pattern)
{
    uint64_t m = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) m |= (uint64_t)flags_to_nibble(zero_byte_flags(d[i] ^ pattern)) << (4 * i);
    return m;
}

// masks of the 64 bytes at file offset `off` (the caller guarantees 16-byte alignment of base)
__device__ __forceinline__ ChunkMasks load_chunk(const uint8_t *base, uint64_t off, uint64_t n, uint32_t (&d)[16])
{
    ChunkMasks c;
    const uint64_t lim = (n + 15) & ~(uint64_t)15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint4 v = {0, 0, 0, 0};
        if (off + 16u * j < lim) v = *reinterpret_cast<const uint4 *>(base + off + 16u * j);
        d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w;
    }
    c.in = off >= n ? 0ull : (n - off >= 64 ? ~0ull : ((1ull << (n - off)) - 1ull));
    c.nl = byte_eq_mask64(d, 0x0A0A0A0Au) & c.in;
    c.gt = byte_eq_mask64(d, 0x3E3E3E3Eu) & c.in;
    c.fq = (byte_eq_mask64(d, 0x40404040u) | byte_eq_mask64(d, 0x2B2B2B2Bu)) & c.in;
    // kept by a sequence line: byte > 0x20 and != 0x7F.  byte > 0x20  <=>  (byte & 0xE0) != 0 and byte != 0x20
    uint64_t low = 0, del = 0, sp = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        low |= (uint64_t)flags_to_nibble(zero_byte_flags(d[i] & 0xE0E0E0E0u)) << (4 * i); // byte < 0x20
        sp |= (uint64_t)flags_to_nibble(zero_byte_flags(d[i] ^ 0x20202020u)) << (4 * i);
        del |= (uint64_t)flags_to_nibble(zero_byte_flags(d[i] ^ 0x7F7F7F7Fu)) << (4 * i);
    }
    c.keepc = ~(low | sp | del) & c.in;
    const bool prev_nl = off == 0 ? true : (off <= n && base[off - 1] == '\n');
    c.ls = ((c.nl << 1) | (prev_nl ? 1ull : 0ull)) & c.in;
    return c;
}

// header-line bytes of a chunk whose first byte belongs to a line in state `in_state` (kSeq / kHdr)
__device__ __forceinline__ uint64_t header_mask(const ChunkMasks &c, uint32_t in_state)
{
    uint64_t hdr = 0, rest = c.ls;
    uint64_t from = 0; // start of the current segment (bit index)
    bool cur = in_state == kHdr;
    while (rest) {
        const int q = __builtin_ctzll(rest);
        if (cur && q > (int)from) hdr |= ((q >= 64 ? 0ull : (1ull << q)) - 1ull) & ~((1ull << from) - 1ull);
        cur = (c.gt >> q) & 1ull;
        from = (uint64_t)q;
        rest &= rest - 1ull;
    }
    if (cur) hdr |= ~((1ull << from) - 1ull);
    return hdr & c.in;
}

// state of the line open at the end of the chunk, kPass if no line starts inside it
__device__ __forceinline__ uint32_t chunk_state(const ChunkMasks &c)
{
    if (!c.ls) return kPass;
    const int q = 63 - __builtin_clzll(c.ls);
    return ((c.gt >> q) & 1ull) ? kHdr : kSeq;
}

// exclusive scan of per-thread states with `combine`, and of two counters; one barrier each
__device__ __forceinline__ uint32_t block_scan_state(uint32_t s, uint32_t *tmp, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl = combine(v, incl);
    }
    uint32_t excl = __shfl_up(incl, 1);
    if (lane == 0) excl = kPass;
    if (lane == 63) tmp[wave] = incl;
    __syncthreads();
    uint32_t before = kPass, all = kPass;
    for (int w = 0; w < kFaThreads / 64; ++w) {
        if (w < wave) before = combine(before, tmp[w]);
        all = combine(all, tmp[w]);
    }
    total = all;
    return combine(before, excl);
}
__device__ __forceinline__ uint32_t block_scan_add(uint32_t v, uint32_t *tmp, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) tmp[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < kFaThreads / 64; ++w) {
        if (w < wave) before += tmp[w];
        all += tmp[w];
    }
    total = all;
    return before + incl - v;
}

// bytes of the chunk that go to the output for a given start state: the sequence-line bytes that are not blank, and the
// newline that ends a header line
__device__ __forceinline__ uint64_t kept_mask(const ChunkMasks &c, uint32_t in_state)
{
    const uint64_t hdr = header_mask(c, in_state);
    return (~hdr & c.keepc) | (hdr & c.nl);
}

} // namespace

// summary of tile t: [0] state at its end (kPass: no line starts inside), [1] kept bytes if it starts inside a sequence
// line, [2] kept bytes if it starts inside a header line
__global__ __launch_bounds__(kFaThreads) void fasta_scan_kernel(const uint8_t *base, uint64_t n, uint32_t *summary, uint32_t *flags)
{
    __shared__ uint32_t tmp[8];
    const uint64_t off = (uint64_t)blockIdx.x * kFaTile + (uint64_t)threadIdx.x * kFaBytesPerThread;
    uint32_t d[16];
    const ChunkMasks c = load_chunk(base, off, n, d);
    if (c.ls & c.fq) atomicOr(flags, 1u); // a line that starts with '@' or '+': FASTQ syntax, not for this path
    uint32_t tile_state;
    const uint32_t st_in = block_scan_state(chunk_state(c), tmp, tile_state);
    // threads whose start state is known count once; the others (in front of the tile's first line start) count twice
    uint32_t known = 0, if_seq = 0, if_hdr = 0;
    if (st_in != kPass) known = (uint32_t)__builtin_popcountll(kept_mask(c, st_in));
    else {
        if_seq = (uint32_t)__builtin_popcountll(kept_mask(c, kSeq));
        if_hdr = (uint32_t)__builtin_popcountll(kept_mask(c, kHdr));
    }
    __syncthreads();
    uint32_t t_known, t_seq, t_hdr;
    block_scan_add(known, tmp, t_known);
    __syncthreads();
    block_scan_add(if_seq, tmp, t_seq);
    __syncthreads();
    block_scan_add(if_hdr, tmp, t_hdr);
    if (threadIdx.x == 0) {
        summary[3 * blockIdx.x] = tile_state;
        summary[3 * blockIdx.x + 1] = t_known + t_seq;
        summary[3 * blockIdx.x + 2] = t_known + t_hdr;
    }
}

// One workgroup: start state and output offset of every tile.  tile_in[t] = state tile t starts in, tile_off[t] = its first
// output byte; tile_off[ntiles] = total output size.
__global__ __launch_bounds__(1024) void fasta_offsets_kernel(const uint32_t *summary, uint32_t ntiles, uint32_t *tile_in, uint64_t *tile_off)
{
    __shared__ uint32_t st[1024];
    __shared__ unsigned long long sums[1024];
    const uint32_t t = threadIdx.x, per = (ntiles + 1023) / 1024;
    const uint32_t lo = min(t * per, ntiles), hi = min(lo + per, ntiles);
    uint32_t s = kPass;
    for (uint32_t i = lo; i < hi; ++i) s = combine(s, summary[3 * i]);
    st[t] = s;
    __syncthreads();
    // the file starts with a header line ('>' checked by the host), so the stream state in front of tile 0 is irrelevant:
    // tile 0 has a line start at byte 0.  Serial combine over 1024 partials by every thread (1024 LDS reads, cheap enough).
    uint32_t in = kSeq;
    for (uint32_t i = 0; i < t; ++i) in = combine(in, st[i]);
    unsigned long long sum = 0;
    uint32_t cur = in;
    for (uint32_t i = lo; i < hi; ++i) {
        tile_in[i] = cur;
        sum += cur == kHdr ? summary[3 * i + 2] : summary[3 * i + 1];
        cur = combine(cur, summary[3 * i]);
    }
    sums[t] = sum;
    __syncthreads();
    unsigned long long before = 0;
    for (uint32_t i = 0; i < t; ++i) before += sums[i];
    cur = in;
    for (uint32_t i = lo; i < hi; ++i) {
        tile_off[i] = before;
        before += cur == kHdr ? summary[3 * i + 2] : summary[3 * i + 1];
        cur = combine(cur, summary[3 * i]);
    }
    if (t == 1023) tile_off[ntiles] = before;
}

// writes the kept bytes of tile t at out + tile_off[t]; every record separator (the newline that ends a header line) is
// also noted in seps[] (output position; unordered, at most seps_cap of them are stored, *nseps counts all)
__global__ __launch_bounds__(kFaThreads) void fasta_compact_kernel(const uint8_t *base, uint64_t n, const uint32_t *tile_in, const uint64_t *tile_off,
                                                                    uint8_t *out, uint64_t *seps, uint32_t seps_cap, uint32_t *nseps)
{
    __shared__ uint32_t tmp[8];
    const uint64_t off = (uint64_t)blockIdx.x * kFaTile + (uint64_t)threadIdx.x * kFaBytesPerThread;
    uint32_t d[16];
    const ChunkMasks c = load_chunk(base, off, n, d);
    uint32_t tile_state;
    uint32_t st_in = block_scan_state(chunk_state(c), tmp, tile_state);
    if (st_in == kPass) st_in = tile_in[blockIdx.x];
    const uint64_t hdr = header_mask(c, st_in);
    uint64_t keep = (~hdr & c.keepc) | (hdr & c.nl);
    __syncthreads();
    uint32_t total;
    const uint32_t before = block_scan_add((uint32_t)__builtin_popcountll(keep), tmp, total);
    uint64_t pos = tile_off[blockIdx.x] + before;
    const uint64_t sepbits = hdr & c.nl;
    while (keep) {
        const int q = __builtin_ctzll(keep);
        out[pos] = base[off + q]; // from the cache: a register array indexed at run time would live in scratch
        if ((sepbits >> q) & 1ull) {
            const uint32_t i = atomicAdd(nseps, 1u);
            if (i < seps_cap) seps[i] = pos;
        }
        ++pos;
        keep &= keep - 1ull;
    }
}

size_t fasta_workspace_bytes(uint64_t n, size_t *o_summary, size_t *o_in, size_t *o_off, size_t *o_flags)
{
    const uint64_t ntiles = (n + kFaTile - 1) / kFaTile;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    *o_summary = o; o += up((size_t)ntiles * 3 * 4);
    *o_in = o; o += up((size_t)ntiles * 4);
    *o_off = o; o += up((size_t)(ntiles + 1) * 8);
    *o_flags = o; o += 256; // [0] format flags, [1] separator count
    return o;
}

hipError_t launch_fasta_compact(const uint8_t *base, uint64_t n, uint8_t *ws, uint8_t *out, uint64_t *seps, uint32_t seps_cap, hipStream_t st)
{
    size_t os, oi, oo, of;
    fasta_workspace_bytes(n, &os, &oi, &oo, &of);
    const uint32_t ntiles = (uint32_t)((n + kFaTile - 1) / kFaTile);
    uint32_t *summary = (uint32_t *)(ws + os), *tin = (uint32_t *)(ws + oi), *flags = (uint32_t *)(ws + of);
    uint64_t *toff = (uint64_t *)(ws + oo);
    hipError_t e = hipMemsetAsync(flags, 0, 256, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fasta_scan_kernel, dim3(ntiles), dim3(kFaThreads), 0, st, base, n, summary, flags);
    hipLaunchKernelGGL(fasta_offsets_kernel, dim3(1), dim3(1024), 0, st, summary, ntiles, tin, toff);
    hipLaunchKernelGGL(fasta_compact_kernel, dim3(ntiles), dim3(kFaThreads), 0, st, base, n, tin, toff, out, seps, seps_cap, flags + 1);
    return hipGetLastError();
}

} // namespace mhx
