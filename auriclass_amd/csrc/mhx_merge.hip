// mhx_merge.hip -- the sharded path's merge where the data is (north_star: "RCCL all-gather of per-shard
// partial bottom-s heaps over xGMI before the final merge").  After the all-gather every rank holds all ranks' slabs in
// HBM: `nranks` unsorted lists of (hash, count), every hash <= its shard's threshold, ~1.1 s entries each without a
// multiplicity filter and 10-14 s with one (every singleton below the threshold travels: counts must stay summable).
// The union's sketch = the s smallest hashes <= T_min whose counts, summed over the ranks, reach m.
//
// Hash values are uniform, so cutting [0, T_min] into equal value bins cuts the union into equal work:
//   merge_scatter_kernel  every workgroup takes a chunk of one slab, counts its entries per bin in LDS, reserves room
//                         in each bin's region with ONE global atomic per (workgroup, bin) and writes the entries there
//                         (a global atomic per entry would serialise on the memory-side atomic units: ~2.7 G/s)
//   merge_bin_kernel      one workgroup per bin: the bin's ~1500 entries go through an LDS hash table (64-bit CAS on the
//                         key, atomic add on the count), the entries with count >= m are ranked among themselves and
//                         written back, in order, to the head of the bin's region
//   merge_compact_kernel  prefix over the bins' counts -> the qualifying entries, globally ascending, into the
//                         pinned result block of the sketcher, header last
// No pass over the 200 MB candidate table, no atomics on it, nothing sorted on the host.  A bin region or table that
// overflows (non-uniform input) raises a flag and the caller falls back to the table path (slab_insert_kernel).
#include "mhx_device.h"

namespace mhx {

// slab entries per workgroup of the scatter pass: small enough that a merge of a few million entries fills the chip (65536,
// the first choice, gave 8 x 704 k entries 88 workgroups on 256 CUs: 0.31 ms), large enough that the one global atomic
// per (workgroup, bin) is shared by a few entries
constexpr uint32_t kMergeChunkDefault = 8192;
constexpr uint32_t kMergeMaxQual = 1024; // qualifying entries a bin can rank in LDS

constexpr int kScatterThreads = 1024, kScatterBatch = 4; // loads in flight per thread: the passes are chains of HBM round trips otherwise

__global__ __launch_bounds__(kScatterThreads) void merge_scatter_kernel(const MergeArgs a)
{
    extern __shared__ uint32_t smem[]; // [nbins] counts, then local cursors | [nbins] bases
    uint32_t *cnt = smem, *base = smem + a.nbins;
    const uint32_t r = blockIdx.y;
    const uint64_t n = a.n[r];
    const uint64_t i0 = (uint64_t)blockIdx.x * a.chunk;
    if (i0 >= n) return;
    const uint64_t i1 = i0 + a.chunk < n ? i0 + a.chunk : n;
    const uint64_t *hashes = a.slabs + (uint64_t)r * a.slab_words + a.hdr_words;
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(hashes + a.cap);
    for (uint32_t b = threadIdx.x; b < a.nbins; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    constexpr uint64_t kStep = (uint64_t)kScatterThreads * kScatterBatch;
    for (uint64_t j = i0 + threadIdx.x; j < i1; j += kStep) {
        uint64_t h[kScatterBatch];
#pragma unroll
        for (int u = 0; u < kScatterBatch; ++u) {
            const uint64_t i = j + (uint64_t)u * kScatterThreads;
            h[u] = i < i1 ? hashes[i] : kEmptyKey;
        }
#pragma unroll
        for (int u = 0; u < kScatterBatch; ++u)
            if (h[u] <= a.t_min && h[u] != kEmptyKey) atomicAdd(&cnt[(uint32_t)(h[u] >> a.shift)], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < a.nbins; b += blockDim.x) {
        const uint32_t c = cnt[b];
        base[b] = c ? atomicAdd(&a.cursor[b], c) : 0u;
        cnt[b] = 0;
    }
    __syncthreads();
    bool over = false;
    for (uint64_t j = i0 + threadIdx.x; j < i1; j += kStep) {
        uint64_t h[kScatterBatch];
        uint32_t c[kScatterBatch];
#pragma unroll
        for (int u = 0; u < kScatterBatch; ++u) {
            const uint64_t i = j + (uint64_t)u * kScatterThreads;
            h[u] = i < i1 ? hashes[i] : kEmptyKey;
            c[u] = i < i1 ? counts[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kScatterBatch; ++u) {
            if (h[u] > a.t_min || h[u] == kEmptyKey) continue;
            const uint32_t b = (uint32_t)(h[u] >> a.shift);
            const uint32_t pos = base[b] + atomicAdd(&cnt[b], 1u);
            if (pos < a.region) {
                a.sc_keys[(uint64_t)b * a.region + pos] = h[u];
                a.sc_cnts[(uint64_t)b * a.region + pos] = c[u];
            } else {
                over = true;
            }
        }
    }
    if (over) atomicOr(a.flags, 1u);
}

__global__ __launch_bounds__(256) void merge_bin_kernel(const MergeArgs a)
{
    extern __shared__ unsigned long long lds[]; // [slots] keys | [slots] u32 counts | [kMergeMaxQual] keys | [kMergeMaxQual] u32 counts
    unsigned long long *keys = lds;
    uint32_t *cnts = reinterpret_cast<uint32_t *>(keys + a.table_slots);
    unsigned long long *qk = reinterpret_cast<unsigned long long *>(cnts + a.table_slots);
    uint32_t *qc = reinterpret_cast<uint32_t *>(qk + kMergeMaxQual);
    __shared__ uint32_t nq;
    const uint32_t b = blockIdx.x, mask = a.table_slots - 1;
    const uint32_t filled = a.cursor[b];
    const uint32_t n = filled < a.region ? filled : a.region;
    for (uint32_t i = threadIdx.x; i < a.table_slots; i += blockDim.x) { keys[i] = kEmptyKey; cnts[i] = 0; }
    if (threadIdx.x == 0) nq = 0;
    __syncthreads();
    if (n > (a.table_slots * 3u) / 4u) { // cannot happen with region <= 3/4 of the table; kept as a guard
        if (threadIdx.x == 0) { atomicOr(a.flags, 2u); a.qn[b] = 0; a.cursor[b] = 0; }
        return;
    }
    const uint64_t *rk = a.sc_keys + (uint64_t)b * a.region;
    const uint32_t *rc = a.sc_cnts + (uint64_t)b * a.region;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t h = rk[i];
        const uint32_t c = rc[i];
        // the bits below the bin index tell the entries of one bin apart
        uint32_t sl = (uint32_t)((h * 0x9E3779B97F4A7C15ull) >> 40) & mask;
        for (;;) {
            unsigned long long cur = keys[sl];
            if (cur == kEmptyKey) cur = atomicCAS(&keys[sl], (unsigned long long)kEmptyKey, (unsigned long long)h);
            if (cur == kEmptyKey || cur == h) { atomicAdd(&cnts[sl], c); break; }
            sl = (sl + 1) & mask;
        }
    }
    __syncthreads();
    bool over = false;
    for (uint32_t i = threadIdx.x; i < a.table_slots; i += blockDim.x) {
        if (keys[i] != kEmptyKey && cnts[i] >= a.min_mult) {
            const uint32_t p = atomicAdd(&nq, 1u);
            if (p < kMergeMaxQual) { qk[p] = keys[i]; qc[p] = cnts[i]; }
            else over = true;
        }
    }
    if (over) atomicOr(a.flags, 4u);
    __syncthreads();
    const uint32_t q = nq < kMergeMaxQual ? nq : kMergeMaxQual;
    // in order, back to the head of the bin's own region (its inputs are all in LDS by now)
    uint64_t *ok = a.sc_keys + (uint64_t)b * a.region;
    uint32_t *oc = a.sc_cnts + (uint64_t)b * a.region;
    for (uint32_t t = threadIdx.x; t < q; t += blockDim.x) {
        const unsigned long long mine = qk[t];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < q; ++j) rank += qk[j] < mine ? 1u : 0u; // the keys of a table are distinct
        if (rank < a.region) { ok[rank] = mine; oc[rank] = qc[t]; }
    }
    if (threadIdx.x == 0) {
        a.qn[b] = q < a.region ? q : a.region;
        a.cursor[b] = 0; // zero again for the next merge
    }
}

// out: the sketcher's pinned result block [n, T, flags, 0 | hashes[out_cap] | counts[out_cap]] (the layout finish() reads)
constexpr int kCompactThreads = 1024;
__global__ __launch_bounds__(kCompactThreads) void merge_compact_kernel(const MergeArgs a, uint64_t *out, uint32_t out_cap)
{
    constexpr int kWaves = kCompactThreads / 64;
    __shared__ uint32_t wave_sums[4];
    __shared__ uint32_t red[2][kWaves];
    __shared__ uint32_t block_base, grand_total;
    __shared__ uint32_t s_off[257];
    // every workgroup owns 256 bins and sums the counts of the bins in front of its own
    const uint32_t first = blockIdx.x * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t before = 0, all = 0;
    for (uint32_t b = threadIdx.x; b < a.nbins; b += kCompactThreads) {
        const uint32_t v = a.qn[b];
        all += v;
        if (b < first) before += v;
    }
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
    if (lane == 0) { red[0][wave] = before; red[1][wave] = all; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t x = 0, y = 0;
        for (int w = 0; w < kWaves; ++w) { x += red[0][w]; y += red[1][w]; }
        block_base = x;
        grand_total = y;
    }
    // exclusive scan of the 256 bins' counts (threads 0 .. 255, one bin each)
    const uint32_t b = first + threadIdx.x;
    const uint32_t mine = threadIdx.x < 256 && b < a.nbins ? a.qn[b] : 0u;
    uint32_t v = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    if (lane == 63 && wave < 4) wave_sums[wave] = v;
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t off = block_base + v - mine;
        for (int w = 0; w < wave; ++w) off += wave_sums[w];
        s_off[threadIdx.x] = off;
        if (threadIdx.x == 255) s_off[256] = off + mine;
    }
    __syncthreads();
    // The workgroup copies its bins' entries TOGETHER, entry e by thread e mod 1024, two entries in flight per thread
    // (coalesced within a bin's run).  Only the first workgroups have anything to copy -- the block holds the first ~s
    // entries -- so their threads must be many and their loads independent: a thread per bin walking its ~100 entries
    // one dependent load after the other took 0.14 ms at 8 x 704 k entries, 256 threads with one entry in flight 0.12.
    uint32_t *oc = reinterpret_cast<uint32_t *>(out + 4 + out_cap);
    const uint32_t lo = s_off[0], hi = s_off[256] < out_cap ? s_off[256] : out_cap;
    auto source = [&](uint32_t e) { // the bin of entry e: the last i with s_off[i] <= e (empty bins share an offset with their successor)
        uint32_t x = 0, y = 256;
        while (y - x > 1) {
            const uint32_t mid = (x + y) >> 1;
            if (s_off[mid] <= e) x = mid; else y = mid;
        }
        return (uint64_t)(first + x) * a.region + (e - s_off[x]);
    };
    for (uint32_t e = lo + threadIdx.x; e < hi; e += 2 * kCompactThreads) {
        const uint32_t e2 = e + kCompactThreads;
        const bool two = e2 < hi;
        const uint64_t s1 = source(e), s2 = two ? source(e2) : s1;
        const uint64_t k1 = a.sc_keys[s1], k2 = a.sc_keys[s2];
        const uint32_t c1 = a.sc_cnts[s1], c2 = a.sc_cnts[s2];
        out[4 + e] = k1;
        oc[e] = c1;
        if (two) { out[4 + e2] = k2; oc[e2] = c2; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = grand_total;
        out[1] = a.t_min;
        out[2] = (uint64_t)*a.flags;
        out[3] = 0;
        *a.flags = 0;
    }
}

hipError_t launch_merge_bins(const MergeArgs &a_in, uint64_t max_n, uint64_t *out, uint32_t out_cap, hipStream_t st)
{
    MergeArgs a = a_in;
    if (a.nbins < 256 || a.nbins > kMergeMaxBins || (a.nbins & (a.nbins - 1)) || a.table_slots < 256 || a.table_slots > kMergeMaxSlots ||
        (a.table_slots & (a.table_slots - 1)))
        return hipErrorInvalidValue;
    static const uint32_t chunk_knob = getenv("MHX_MERGE_CHUNK") ? (uint32_t)atol(getenv("MHX_MERGE_CHUNK")) : 0u; // experiment knob
    a.chunk = chunk_knob >= 1024 ? chunk_knob : kMergeChunkDefault;
    const unsigned chunks = (unsigned)((max_n + a.chunk - 1) / a.chunk);
    const size_t scatter_lds = 2 * (size_t)a.nbins * sizeof(uint32_t);
    const size_t bin_lds = (size_t)a.table_slots * 12 + (size_t)kMergeMaxQual * 12;
    static bool attr_set = false; // dynamic LDS beyond 64 KB has to be asked for once per kernel
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(merge_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kMergeMaxBins * 4);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(merge_bin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMergeMaxSlots * 12 + kMergeMaxQual * 12);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (chunks) hipLaunchKernelGGL(merge_scatter_kernel, dim3(chunks, a.nranks), dim3(kScatterThreads), scatter_lds, st, a);
    hipLaunchKernelGGL(merge_bin_kernel, dim3(a.nbins), dim3(256), bin_lds, st, a);
    hipLaunchKernelGGL(merge_compact_kernel, dim3(a.nbins / 256), dim3(kCompactThreads), 0, st, a, out, out_cap);
    return hipGetLastError();
}

} // namespace mhx
