// mhx_kernels.hip -- gfx950 kernels of the sketch + distance engine.
//
//   sketch_tile_kernel<K,FMT>  one workgroup per 16 KiB tile of the FASTQ / sequence byte
//                              stream: stage -> classify -> (FASTQ) line phase -> candidate k-mer
//                              starts -> LDS work list -> canonical k-mer + MurmurHash3_x64_128 ->
//                              admission -> table.  FMT 0: sequence stream; FMT 2: FASTQ, every tile
//                              finds its line phase by itself (no ticket, no inter-workgroup wait);
//                              FMT 1: FASTQ with ticket + decoupled look-back, the repair pass for the
//                              tiles FMT 2 had to leave out (lines too long to self-synchronise)
//   table_tighten              tightens the admission threshold T from the candidate table (one launch)
//   table_reset                vacates the table and clears the control buffers (one launch)
//   table_extract              compact (hash,count) entries <= limit for the host / all-gather
//   dist_pairs_kernel          mash compareSketches for a batch of (query, ref) pairs
//
// Replaces the loops inside the `mash sketch` / `mash dist` child processes that
// /root/reference/auriclass/classes.py:576-596, 696-713 and 92-104 spawn.
#include "mhx_device.h"
#include "mhx_tile.h"

namespace mhx {

// ---------------------------------------------------------------------------------------
// candidate table: open addressing, keys claimed with a 64-bit CAS, counts by atomic add.
// Only atomics touch the table inside a launch, so no cross-XCD visibility protocol is
// needed; the next kernel on the stream reads it with plain loads.
// ---------------------------------------------------------------------------------------
struct DeviceInserter {
    unsigned long long *keys;
    uint32_t *cnts;
    uint64_t mask;
    unsigned long long *stats; // this block's replica
    __device__ __forceinline__ void operator()(uint64_t h)
    {
        if (h == kEmptyKey) {
            atomicAdd(&stats[kStatMaxKey], 1ull);
            return;
        }
        uint64_t slot = h & mask;
        for (int probe = 0; probe < 8192; ++probe) {
            const unsigned long long prev = atomicCAS(&keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)h);
            if (prev == kEmptyKey || prev == h) {
                atomicAdd(&cnts[slot], 1u);
                return;
            }
            slot = (slot + 1) & mask;
        }
        atomicOr(&stats[kStatFlags], (unsigned long long)kFlagTableFull);
    }
};

// Not inlined on purpose: the window finished here needs ~40 registers of its own (two strands, eight words each, and
// the hash); inside the kernel body they would be the hash loop's problem.
template <int K> __device__ __noinline__ uint32_t finish_candidate(const TileSmem &sm, uint32_t code, uint64_t T, DeviceInserter ins)
{
    return process_deferred<K>(sm, code, T, ins);
}

// What the hash loop does with a candidate window in the queue form: it appends (group << 3) | window to the free tail
// of the tile's work list (TileSmem) -- one LDS atomic and one 16-bit store, nothing else: no call, no threshold, no table
// pointers in the hot loop's live set (round 2 finished an overflowing candidate on the spot through a non-inlined call,
// which cost the kernel 32 bytes of scratch per lane for the registers saved around it).  A queue that overflows (more
// than ~1000 candidates in a tile: only while T still admits a large share of all hashes, i.e. the first launches of a
// sketch) is not used at all: the counter says so and the tile then finishes EVERY valid window through the generic
// routine after the hash loop (sketch_tile_kernel), which tests each hash against T itself.
struct CandidateQueue {
    TileSmem &sm;
    uint32_t first, cap; // list[first .. first + cap) is free
    __device__ __forceinline__ void operator()(uint32_t group, int window) const
    {
        const uint32_t slot = atomicAdd(&sm.misc[7], 1u);
        if (slot < cap) sm.list[first + slot] = (uint16_t)((group << 3) | (uint32_t)window);
    }
};

// ---------------------------------------------------------------------------------------
// Decoupled look-back over per-tile newline counts (wave 0 of the block).
// tile_state[t] is ONE naturally aligned 8-byte word {flag:32 | value:32} written by one
// agent-scope store and polled by agent-scope loads: flag 1 = this tile's own count,
// flag 2 = inclusive prefix up to and including this tile.  Tiles are handed out by an
// atomic ticket, so every predecessor of a running tile is itself running or done and
// publishes its own count without waiting on anyone: the wait below always ends.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ld_state(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ uint32_t lookback_wave(uint64_t *state, uint32_t tile, uint32_t first_tile, uint32_t agg,
                                  unsigned long long *stats)
{
    const int lane = threadIdx.x & 63;
    constexpr uint64_t A = 1ull << 32, P = 2ull << 32;
    if (tile == first_tile) {
        if (lane == 0) st_state(&state[tile], P | agg);
        return 0;
    }
    if (lane == 0) st_state(&state[tile], A | agg);
    uint32_t excl = 0;
    int64_t pos = (int64_t)tile - 1;
    // Every round polls the 4 x 64 nearest unread predecessors with four loads in flight at once:
    // tiles reach this point faster than one L2 round trip per 64 of them, so a 64-wide window
    // per round trip never catches up with the newest published prefix.
    constexpr int kWin = 4;
    bool done = false;
    for (uint32_t spins = 0; !done;) {
        uint64_t w[kWin];
#pragma unroll
        for (int j = 0; j < kWin; ++j) {
            const int64_t idx = pos - lane - 64 * j;
            w[j] = idx >= (int64_t)first_tile ? ld_state(&state[idx]) : P; // before the push: prefix 0
        }
        bool blocked = false;
#pragma unroll
        for (int j = 0; j < kWin; ++j) {
            if (done || blocked) continue;
            const uint32_t f = (uint32_t)(w[j] >> 32);
            const uint64_t notready = __ballot(f == 0);
            const uint64_t isp = __ballot(f == 2);
            uint64_t take = 0; // lanes whose value is added
            if (isp) {
                const int q = __builtin_ctzll(isp);
                const uint64_t below = q ? (~0ull >> (64 - q)) : 0ull;
                if ((notready & below) == 0) {
                    take = below | (1ull << q);
                    done = true;
                }
            } else if (!notready) {
                take = ~0ull;
            }
            if (take) {
                uint32_t v = ((take >> lane) & 1ull) ? (uint32_t)w[j] : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                excl += v;
                if (!done) pos -= 64;
            } else {
                blocked = true;
            }
        }
        if (blocked) {
            if (++spins > (1u << 20)) { // ~a second; never reached in a healthy launch
                if (lane == 0) atomicOr(&stats[kStatFlags], (unsigned long long)kFlagSpinTimeout);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    if (lane == 0) st_state(&state[tile], P | (uint64_t)(excl + agg));
    return excl;
}

// exclusive prefix sum over the workgroup's threads (wave shuffles + 4 partials through LDS);
// every thread also gets the workgroup total.  One barrier.
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t value, uint32_t *wave_sums, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v = value;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    if (lane == 63) wave_sums[wave] = v;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        const uint32_t x = wave_sums[w];
        if (w < wave) base += x;
        all += x;
    }
    total = all;
    return base + v - value;
}

// ---------------------------------------------------------------------------------------
#ifndef MHX_MIN_WAVES
#define MHX_MIN_WAVES 7   // 72 VGPRs: seven waves per SIMD, matching the seven workgroups per CU the LDS footprint admits
#endif
template <int K, int FMT, bool QUEUE> __global__ __launch_bounds__(kBlock, MHX_MIN_WAVES) void sketch_tile_kernel(const HashArgs a)
{
    __shared__ TileSmem sm;
    constexpr bool FASTQ = (FMT != 0), LOOKBACK = (FMT == 1), SELFSYNC = (FMT == 2);
    const int tid = threadIdx.x;
    // the parsing phases are latency-bound (loads, barriers, look-back): at high priority their few instructions do
    // not queue behind the hash loops of the other resident workgroups, so a wave reaches its own hash loop sooner
    // and more waves per SIMD are in VALU-dense code at any time (-2 % kernel time)
    __builtin_amdgcn_s_setprio(3);

    // tile id: in ticket order for FASTQ (look-back needs started-before ordering)
    uint32_t tile;
    if (LOOKBACK) {
        if (tid == 0) sm.misc[2] = a.tile0 + atomicAdd(a.ticket, 1u);
        __syncthreads();
        tile = sm.misc[2];
    } else {
        tile = a.tile0 + blockIdx.x;
    }
    const uint64_t tile_off = (uint64_t)tile * kTileBytes;
    unsigned long long *stats = reinterpret_cast<unsigned long long *>(a.stats) + (tile % kStatReplicas) * kStatCount;
    if (tid == 0) { sm.misc[3] = 0; sm.misc[4] = 0; sm.misc[5] = 0; sm.misc[7] = 0; }
#ifdef MHX_STAMPS
    uint64_t stamp_prev = clock64();
    int stamp_idx = 0;
#define MHX_STAMP()                                                                                      \
    do {                                                                                                 \
        if (tid == 0) {                                                                                  \
            const uint64_t now_ = clock64();                                                             \
            atomicAdd(&stats[kStatStamp0 + stamp_idx], (unsigned long long)(now_ - stamp_prev));         \
            stamp_prev = now_;                                                                           \
        }                                                                                                \
        ++stamp_idx;                                                                                     \
    } while (0)
#else
#define MHX_STAMP() do { } while (0)
#endif

    phase_stage(sm, tid, a.base, tile_off, a.end);
    __syncthreads();
    MHX_STAMP(); // 0: stage (global loads -> LDS)

    ThreadState st;
    const bool interior = tile_off >= a.begin && tile_off + kTileBytes + kHaloBytes <= a.end; // nothing to mask out
    phase_classify(sm, tid, st, tile_off, a.begin, a.end, interior);

    MHX_STAMP(); // 1: classify
    uint32_t line_base = 0, excl = 0, tile_total = 0;
    // the format look-ahead may only read staged bytes that belong to the span
    const uint64_t span_left = a.end > tile_off ? a.end - tile_off : 0;
    const uint32_t check_limit = span_left < (uint64_t)(kTileBytes + kHaloBytes) ? (uint32_t)span_left : (uint32_t)(kTileBytes + kHaloBytes);
    if (FASTQ) {
        excl = block_scan_excl(st.nlcount, sm.cnt, tile_total); // barrier inside: the newline map is complete
        // the tile that holds the start of the span begins a record there; every other tile looks at its own first lines
        if (tid == 0) sm.misc[0] = tile == a.first_tile ? 0u : phase_selfsync(sm, check_limit);
        __syncthreads();
        const uint32_t self_phase = sm.misc[0];
        if (SELFSYNC) {
            if (self_phase == 4u) { // lines too long to tell: left to the look-back pass, nothing of this tile is counted now
                if (tid == 0) { a.phase_rec[tile - a.first_tile] = 0; atomicOr(a.need_lookback, 1u); }
                return;
            }
            if (tid == 0) a.phase_rec[tile - a.first_tile] = phase_record(self_phase, tile_total);
            line_base = self_phase;
        } else {
            if (a.repair && self_phase != 4u) { // repair pass: this tile was done in the first pass, it only publishes its phase
                if (tid == 0) {
                    st_state(&a.tile_state[tile], (2ull << 32) | (uint64_t)(self_phase + tile_total));
                    a.phase_rec[tile - a.first_tile] = phase_record(self_phase, tile_total);
                }
                return;
            }
            if (tid < 64) {
                const uint32_t lb = lookback_wave(a.tile_state, tile, a.first_tile, tile_total, stats);
                if (tid == 0) sm.misc[0] = lb;
            }
            __syncthreads();
            line_base = sm.misc[0];
            if (a.repair && tid == 0) a.phase_rec[tile - a.first_tile] = phase_record(line_base, tile_total);
        }
    }
    MHX_STAMP(); // 2: newline scan + look-back
    bool bad = false;
    const uint32_t long_records = phase_good<FASTQ>(sm, tid, st, line_base, excl, tile_total, check_limit, bad, tile_off, a.end, (uint32_t)K);
    if (FASTQ && bad) atomicOr(&stats[kStatFlags], (unsigned long long)kFlagBadFastq);
    if (FASTQ && long_records) atomicAdd(&sm.misc[5], long_records);
    __syncthreads();
    MHX_STAMP(); // 3: good-base map

    uint32_t items = 0;
    const uint32_t kmers = phase_runs<K>(sm, tid, items);
    if (kmers) atomicAdd(&sm.misc[3], kmers);
    uint32_t nitems = 0;
    const uint32_t items_before = block_scan_excl(items, sm.cnt + 8, nitems); // barrier inside: sm.valid complete
    phase_compact(sm, tid, items_before);
    __syncthreads();
    MHX_STAMP(); // 4: valid starts + work list
    __builtin_amdgcn_s_setprio(0);

    const uint64_t T = *a.thresh;
    const uint32_t limit = admission_limit(T);
    DeviceInserter ins{reinterpret_cast<unsigned long long *>(a.keys), a.cnts, a.slot_mask, stats};
    const uint32_t qcap = (uint32_t)kGroupsPerTile - nitems; // QUEUE: the work list's unused tail holds the candidate queue
    CandidateQueue queue{sm, nitems, qcap};
    uint32_t ninsert = 0;
    for (uint32_t it = tid; it < nitems; it += kBlock) ninsert += process_group<K, QUEUE>(sm, sm.list[it], T, limit, ins, queue);
    if (QUEUE) {
        __syncthreads();
        const uint32_t ncand = sm.misc[7];
        if (ncand <= qcap) { // the candidates the loop has queued, one per lane
            for (uint32_t c = tid; c < ncand; c += kBlock) ninsert += finish_candidate<K>(sm, sm.list[nitems + c], T, ins);
        } else { // the queue overflowed (T still admits a large share of all hashes): every valid window, whole hash, exact test
            for (uint32_t it = tid; it < nitems; it += kBlock) {
                const uint32_t g = sm.list[it];
                uint32_t vmask = reinterpret_cast<const uint8_t *>(sm.valid)[g];
                while (vmask) {
                    const uint32_t j = (uint32_t)__builtin_ctz(vmask);
                    vmask &= vmask - 1u;
                    ninsert += finish_candidate<K>(sm, (g << 3) | j, T, ins);
                }
            }
        }
    }
    if (ninsert) atomicAdd(&sm.misc[4], ninsert);
    __syncthreads();
    MHX_STAMP(); // 5: work loop
    if (tid == 0) {
        if (sm.misc[3]) atomicAdd(&stats[kStatKmers], (unsigned long long)sm.misc[3]);
        if (sm.misc[4]) atomicAdd(&stats[kStatInserts], (unsigned long long)sm.misc[4]);
        if (FASTQ && tile_total) atomicAdd(&stats[kStatLines], (unsigned long long)tile_total);
        if (FASTQ && sm.misc[5]) atomicAdd(&stats[kStatRecords], (unsigned long long)sm.misc[5]);
    }
}

template <int K> static hipError_t launch_k(int fmt, const HashArgs &a, hipStream_t st)
{
    // a.queue_candidates: large sketches (many windows pass the admission test) finish their candidates after the hash
    // loop, one per lane; small ones where they are found (process_group_regs)
#define MHX_LAUNCH(FMT_) do { if (a.queue_candidates) hipLaunchKernelGGL((sketch_tile_kernel<K, FMT_, true>), dim3(a.ntiles), dim3(kBlock), 0, st, a); \
                              else hipLaunchKernelGGL((sketch_tile_kernel<K, FMT_, false>), dim3(a.ntiles), dim3(kBlock), 0, st, a); } while (0)
    if (fmt == 1) MHX_LAUNCH(1);
    else if (fmt == 2) MHX_LAUNCH(2);
    else MHX_LAUNCH(0);
#undef MHX_LAUNCH
    return hipGetLastError();
}

#ifdef MHX_ONLY_K   // experiment / ISA-study builds: one k-mer size
#define MHX_K_LIST(X) X(MHX_ONLY_K)
#else
#define MHX_K_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
#endif

bool hash_k_supported(int k) { return k >= 1 && k <= 32; }

hipError_t launch_hash(int k, int fmt, const HashArgs &a, hipStream_t st)
{
    if (a.ntiles == 0) return hipSuccess;
    switch (k) {
#define X(KK) case KK: return launch_k<KK>(fmt, a, st);
        MHX_K_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------
// Threshold tightening.  T only ever decreases, and only to a value below which at least
// s entries with count >= m already exist, so every hash of the final sketch stays admitted
// (and therefore fully counted) for the whole run.
// ---------------------------------------------------------------------------------------
// One kernel: every workgroup histograms its share of the table in LDS and flushes it with global atomics; the
// workgroup that finishes LAST (ticket) turns the histogram into the new threshold.  All cross-workgroup data moves
// through memory-side atomics (the flush, the ticket) and agent-scope loads / stores in the last workgroup (which also
// clear the bins for the next pass), with an agent-scope release in front of the ticket, so no cache of another XCD is
// ever trusted.
__global__ __launch_bounds__(256) void table_tighten_kernel(const TableArgs a)
{
    __shared__ uint32_t h[kHistBins];
    __shared__ uint32_t occ_s, solid_s, last_s, cut_s;
    __shared__ uint32_t wave_sums[4];
    __shared__ unsigned long long tot_s[2];
    for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) h[i] = 0;
    if (threadIdx.x == 0) { occ_s = 0; solid_s = 0; cut_s = 0xFFFFFFFFu; tot_s[0] = 0; tot_s[1] = 0; }
    __syncthreads();
    const uint64_t T = *a.thresh;
    const int lz = T ? __builtin_clzll(T) : 63;
    uint32_t occ = 0, solid = 0;
    // sample = 1: every slot; sample = 8: one 256-slot block in eight (slots are hash-addressed, so
    // any fixed subset of blocks is a uniform sample of the entries)
    const uint64_t nblocks256 = a.nslots / 256;
    const uint64_t step = a.sample > 1 ? a.sample : 1;
    // four independent key loads in flight per thread: with one, the pass is a chain of L2/HBM round trips
    const uint64_t stride = (uint64_t)gridDim.x * step;
    for (uint64_t b0 = (uint64_t)blockIdx.x * step; b0 < nblocks256; b0 += 4 * stride) {
        uint64_t key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t b = b0 + u * stride;
            key[u] = b < nblocks256 ? a.keys[b * 256 + threadIdx.x] : kEmptyKey;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (key[u] == kEmptyKey) continue;
            ++occ;
            if (key[u] <= T && a.cnts[(b0 + u * stride) * 256 + threadIdx.x] >= a.min_mult) {
                ++solid;
                atomicAdd(&h[(key[u] << lz) >> (64 - 11)], 1u);
            }
        }
    }
    if (occ) atomicAdd(&occ_s, occ);
    if (solid) atomicAdd(&solid_s, solid);
    __syncthreads();
    for (int i = threadIdx.x; i < kHistBins; i += blockDim.x)
        if (h[i]) atomicAdd(&a.hist[i], h[i]);
    if (threadIdx.x == 0) {
        // 64 replicas, 64 bytes apart: a thousand workgroups adding to ONE word serialise at ~10-20 ns each
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(a.acc) + (blockIdx.x % kAccReplicas) * 8;
        if (occ_s) atomicAdd(&acc[0], (unsigned long long)occ_s);
        if (solid_s) atomicAdd(&acc[1], (unsigned long long)solid_s);
    }
    // every wave waits for its own atomics, the barrier collects the waves, ONE lane fences and takes the ticket
    // (256 threads fencing cost 60 us per pass)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last_s = atomicAdd(a.done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last_s) return;

    // ---- last workgroup: first bin where the cumulative count reaches s ----------------------------------------
    // T only ever decreases, and only to a value below which at least s entries with count >= m already exist, so
    // every hash of the final sketch stays admitted (and therefore fully counted) for the whole run.
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int kPer = kHistBins / 256; // bins per thread, in value order
    uint32_t c[kPer], mine = 0;
    // agent-scope loads + stores: served past the caches, like the look-back words.  All loads first (one round trip for
    // the eight of them instead of eight), then the stores that clear the bins for the next pass.
#pragma unroll
    for (int j = 0; j < kPer; ++j) c[j] = __hip_atomic_load(&a.hist[kPer * t + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        __hip_atomic_store(&a.hist[kPer * t + j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mine += c[j];
    }
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (int w = 0; w < wave; ++w) before += wave_sums[w];
    // sampled pass: counts are ~Binomial(truth, 1/sample); ask for the expected s/sample plus six
    // standard deviations (+16 for small s) so that the sampling error cannot push T below the true
    // s-th qualifying hash (~1e-9 per pass; finish() counts exactly and refuses a short result below a lowered T)
    uint32_t s = a.sketch_size;
    if (a.sample > 1) {
        const float mean = (float)a.sketch_size / (float)a.sample;
        s = (uint32_t)(mean + 6.0f * sqrtf(mean)) + 16u;
    }
    if (before < s && before + mine >= s) { // the cut lies in one of my bins (exactly one thread gets here)
        uint32_t run = before;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (run < s && run + c[j] >= s) cut_s = (uint32_t)(kPer * t + j);
            run += c[j];
        }
    }
    if (t < kAccReplicas) { // this pass's totals
        uint64_t *acc = a.acc + t * 8;
        const uint64_t o = __hip_atomic_load(&acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t so = __hip_atomic_load(&acc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&acc[0], (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&acc[1], (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o) atomicAdd(&tot_s[0], (unsigned long long)o);
        if (so) atomicAdd(&tot_s[1], (unsigned long long)so);
    }
    __syncthreads();
    if (t == 0) {
        const uint64_t occupied = tot_s[0] * (a.sample > 1 ? a.sample : 1), solid = tot_s[1] * (a.sample > 1 ? a.sample : 1);
        a.stats[kStatOccupied] = occupied;
        a.stats[kStatSolid] = solid;
        *a.done = 0;
        const uint32_t cut = cut_s;
        uint64_t now = T;
        bool established = a.min_mult > 1 && a.stats[kStatEstablished];
        if (cut != 0xFFFFFFFFu && lz <= 52) {
            const uint64_t edge = (((uint64_t)cut + 1) << (53 - lz)) - 1; // last value of bin `cut`
            if (edge < T) {
                now = edge;
                if (a.min_mult > 1) { a.stats[kStatEstablished] = 1; established = true; } // from now on T follows the solid hashes: no more caps
            }
        }
        // the byte-count cap in front of the NEXT launch of the same push (cap_threshold_kernel's rule, without its launch)
        if (a.next_cap && !established && !(occupied > 0 && solid * 5 >= occupied) && now > a.next_cap) {
            now = a.next_cap;
            a.stats[kStatBounded] = 1;
        }
        // (atomic: a pass between two launches of a push runs beside the next launch's cap_threshold_kernel, and T must
        // never rise -- a hash that is admitted now must have been admitted on every earlier occurrence)
        if (now < T) atomicMin(reinterpret_cast<unsigned long long *>(a.thresh), (unsigned long long)now);
    }
}

// Cap of the admission threshold that follows the bytes seen (multiplicity filter, before s solid hashes exist):
// T = min(T, cap) -- decided on the device from what the last tighten pass left in the counters, so that the host never
// waits for a round trip: no cap once a pass has lowered T from solid hashes, and none while the table looks like a
// small genome sequenced deeply (a fifth of its entries solid, yet fewer than s of them: such a sketch may need every
// solid hash there is).
__global__ void cap_threshold_kernel(uint64_t *thresh, uint64_t cap, uint64_t *stats)
{
    if (stats[kStatEstablished]) return;
    const uint64_t occupied = stats[kStatOccupied], solid = stats[kStatSolid];
    if (occupied > 0 && solid * 5 >= occupied) return;
    if (atomicMin(reinterpret_cast<unsigned long long *>(thresh), (unsigned long long)cap) > cap) stats[kStatBounded] = 1;
}

// FASTQ, self-synchronising form: do the line phases the tiles found (HashArgs::phase_rec) form one chain?
__global__ void phase_verify_kernel(const uint8_t *rec, uint32_t ntiles, uint64_t *stats)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1u;
    if (i < ntiles && phase_chain_broken(rec[i - 1], rec[i]))
        atomicOr(reinterpret_cast<unsigned long long *>(stats) + kStatFlags, (unsigned long long)kFlagBadFastq);
}

hipError_t launch_phase_verify(const uint8_t *rec, uint32_t ntiles, uint64_t *stats, hipStream_t st)
{
    if (ntiles < 2) return hipSuccess;
    hipLaunchKernelGGL(phase_verify_kernel, dim3((ntiles - 1 + 255) / 256), dim3(256), 0, st, rec, ntiles, stats);
    return hipGetLastError();
}

hipError_t launch_cap_threshold(uint64_t *thresh, uint64_t cap, uint64_t *stats, hipStream_t st)
{
    hipLaunchKernelGGL(cap_threshold_kernel, dim3(1), dim3(1), 0, st, thresh, cap, stats);
    return hipGetLastError();
}

hipError_t launch_tighten(const TableArgs &a, hipStream_t st)
{
    // few, long-running workgroups: each one flushes its 2048-bin LDS histogram with global atomics
    uint64_t blocks = a.nslots / 256 / 16;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(table_tighten_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

// One launch instead of five memsets and a copy: vacates the table and clears every small control buffer.
__global__ __launch_bounds__(256) void table_reset_kernel(const TableArgs a, uint64_t t_init, uint32_t *tickets, uint32_t ntickets, uint32_t *out_n)
{
    const uint64_t n2 = a.nslots / 2, n4 = a.nslots / 4; // nslots is a power of two >= 2^16
    uint4 *k4 = reinterpret_cast<uint4 *>(a.keys), *c4 = reinterpret_cast<uint4 *>(a.cnts);
    const uint4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, zero = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * blockDim.x) {
        k4[i] = ones;
        if (i < n4) c4[i] = zero;
    }
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < (uint32_t)kHistBins; i += blockDim.x) a.hist[i] = 0;
        for (uint32_t i = threadIdx.x; i < (uint32_t)kAccReplicas * 8; i += blockDim.x) a.acc[i] = 0;
        for (uint32_t i = threadIdx.x; i < (uint32_t)(kStatReplicas * kStatCount); i += blockDim.x) a.stats[i] = 0;
        for (uint32_t i = threadIdx.x; i < ntickets; i += blockDim.x) tickets[i] = 0;
        if (threadIdx.x == 0) { *a.thresh = t_init; *a.done = 0; *out_n = 0; if (a.need_lookback) *a.need_lookback = 0; }
    }
}

hipError_t launch_reset(const TableArgs &a, uint64_t t_init, uint32_t *tickets, uint32_t ntickets, uint32_t *out_n, hipStream_t st)
{
    uint64_t blocks = a.nslots / 2 / 256 / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(table_reset_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, t_init, tickets, ntickets, out_n);
    return hipGetLastError();
}

__device__ __forceinline__ int order_shift(uint64_t T, uint32_t log2_buckets)
{ // finish(), large sketches (see launch_order_block): the bucket of a hash is its leading log2_buckets bits below T's top bit
    const int bits = 64 - __clzll((long long)(T | 1ull)); // T < 2^bits
    return bits > (int)log2_buckets ? bits - (int)log2_buckets : 0;
}

__global__ __launch_bounds__(256) void table_extract_kernel(const TableArgs a, uint64_t limit, uint32_t min_count,
                                                            uint64_t *out_keys, uint32_t *out_cnts, uint32_t cap,
                                                            uint32_t *out_n, uint64_t *flags_out, const uint64_t *limit_dev,
                                                            uint64_t *limit_out, uint64_t *maxkey_out,
                                                            uint32_t *order_cursor, uint32_t order_log2,
                                                            uint64_t *hdr_dev, uint64_t *hdr_host, uint32_t *ticket,
                                                            uint64_t *occ_out, uint32_t nhdr, uint64_t *hdr_copy)
{
    if (limit_dev) limit = *limit_dev; // the admission threshold as it stands on the device
    const int bucket_shift = order_shift(limit, order_log2);
    if (blockIdx.x == 0 && threadIdx.x == 0 && limit_out) *limit_out = limit;
    // optional: occurrences of the one hash value the table cannot hold (2^64 - 1), summed over the replicas
    if (maxkey_out && blockIdx.x == 0 && threadIdx.x < kStatReplicas) {
        const uint64_t c = a.stats[threadIdx.x * kStatCount + kStatMaxKey];
        if (c) atomicAdd(reinterpret_cast<unsigned long long *>(maxkey_out), (unsigned long long)c);
    }
    // optional: OR of the replicated device flags, so that a caller that never reads the stats block
    // (the multi-GPU slab export) still learns about a full table or a malformed FASTQ
    if (flags_out && blockIdx.x == 0 && threadIdx.x < kStatReplicas) {
        uint64_t f = a.stats[threadIdx.x * kStatCount + kStatFlags];
        if (threadIdx.x == 0) // plus the state of the m > 1 phase and the "repair pass due" word of the FASTQ parser
            f |= (a.stats[kStatBounded] ? kFlagStateBounded : 0) | (a.stats[kStatEstablished] ? kFlagStateEstablished : 0) |
                 (a.need_lookback && *a.need_lookback ? kFlagNeedLookback : 0);
        if (f) atomicOr(reinterpret_cast<unsigned long long *>(flags_out), (unsigned long long)f);
    }
    // Qualifying entries are sparse (about one per few hundred slots), so they are collected per
    // workgroup in LDS and appended to the output with ONE global atomic per flush instead of one
    // per entry (a single counter word serialises at ~10 ns per atomic).
    constexpr uint32_t kBuf = 1024;
    __shared__ unsigned long long bkeys[kBuf];
    __shared__ uint32_t bcnts[kBuf];
    __shared__ uint32_t nbuf, base, occ_s;
    if (threadIdx.x == 0) { nbuf = 0; occ_s = 0; }
    __syncthreads();
    uint32_t occ = 0; // occupied slots seen by this thread (reported when occ_out is given: the shard export)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (a.nslots + stride - 1) / stride; // same trip count for every thread (barriers inside)
    for (uint64_t rd = 0; rd <= rounds; ++rd) {
        if (rd < rounds) {
            const uint64_t i = rd * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
            if (i < a.nslots) {
                const uint64_t key = a.keys[i];
                occ += key != kEmptyKey ? 1u : 0u;
                if (key != kEmptyKey && key <= limit) {
                    const uint32_t c = a.cnts[i];
                    if (c >= min_count) {
                        const uint32_t p = atomicAdd(&nbuf, 1u); // at most 256 per round, flushed below before it can overflow
                        bkeys[p] = key;
                        bcnts[p] = c;
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t n = nbuf;
        if (n > kBuf - 256 || (rd == rounds && n)) { // flush
            if (threadIdx.x == 0) base = atomicAdd(out_n, n);
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < n; j += blockDim.x)
                if (base + j < cap) {
                    out_keys[base + j] = bkeys[j];
                    out_cnts[base + j] = bcnts[j];
                    if (order_cursor) atomicAdd(&order_cursor[bkeys[j] >> bucket_shift], 1u); // bucket sizes for launch_order_block
                }
            __syncthreads();
            if (threadIdx.x == 0) nbuf = 0;
            __syncthreads();
        }
    }
    // finish(): the four header words [n, T, flags, max-key count] at hdr_dev go to the pinned block (whose payload the
    // flushes above have written directly) and are cleared for the next call, by the workgroup that finishes last --
    // no header memset in front of the kernel, no copy command behind it.  Same hand-over as the tighten pass: every
    // wave waits for its own memory operations, one lane releases and takes the ticket, the last workgroup reads the
    // words with agent-scope loads.
    if (occ_out) { // table occupancy (the sharded merge sizes its insertions against it), one global atomic per workgroup
        if (occ) atomicAdd(&occ_s, occ);
        __syncthreads();
        if (threadIdx.x == 0 && occ_s) atomicAdd(reinterpret_cast<unsigned long long *>(occ_out), (unsigned long long)occ_s);
    }
    if (!hdr_host) return;
    __shared__ uint32_t last_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last_s = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last_s) return;
    if (threadIdx.x < nhdr) {
        const uint64_t v = __hip_atomic_load(&hdr_dev[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hdr_host[threadIdx.x] = v;
        if (hdr_copy) hdr_copy[threadIdx.x] = v; // the shard export: the header also rides in front of the slab
        __hip_atomic_store(&hdr_dev[threadIdx.x], (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) *ticket = 0;
}

hipError_t launch_extract(const TableArgs &a, uint64_t limit, uint32_t min_count, uint64_t *out_keys,
                          uint32_t *out_cnts, uint32_t cap, uint32_t *out_n, uint64_t *flags_out, const uint64_t *limit_dev,
                          uint64_t *limit_out, uint64_t *maxkey_out, hipStream_t st, uint32_t *order_cursor, uint32_t order_log2,
                          uint64_t *hdr_dev, uint64_t *hdr_host, uint32_t *ticket, uint64_t *occ_out, uint32_t nhdr, uint64_t *hdr_copy)
{
    uint64_t blocks = (a.nslots + 256 * 16 - 1) / (256 * 16);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(table_extract_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, limit, min_count, out_keys,
                       out_cnts, cap, out_n, flags_out, limit_dev, limit_out, maxkey_out, order_cursor, order_log2,
                       hdr_dev, hdr_host, ticket, occ_out, nhdr, hdr_copy);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Sharded path, the merge where the data is (mhx_sketcher_merge_slabs): after the all-gather every
// rank holds all ranks' partial results in HBM.  Its own entries already sit in its candidate table with exact
// counts; the other ranks' entries <= T_min are added to that table -- a key is claimed by CAS (or found), its count
// added atomically -- and the ordinary extraction (count >= m, hash <= T_min) then yields the union's sketch.
// One thread per entry, ranks along grid.y; the gathered buffer is `nranks` slabs of `slab_words` 8-byte words:
// hashes[cap], then the u32 counts.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slab_insert_kernel(const SlabMergeArgs a)
{
    const uint32_t r = blockIdx.y;
    if (blockIdx.x == 0 && r == 0 && threadIdx.x == 0) {
        *a.thresh = a.t_min; // what the extraction behind this kernel reads as its limit
        if (a.maxkey_others) atomicAdd(reinterpret_cast<unsigned long long *>(a.stats) + kStatMaxKey, (unsigned long long)a.maxkey_others);
    }
    if (r == a.own_rank) return;
    const uint64_t n = a.n[r];
    const uint64_t *hashes = a.slabs + (uint64_t)r * a.slab_words + a.hdr_words;
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(hashes + a.cap);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(a.keys);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = hashes[i];
        if (h > a.t_min || h == kEmptyKey) continue; // above T_min a shard's list is incomplete: not part of the union's evidence
        const uint32_t c = counts[i];
        uint64_t slot = h & a.slot_mask;
        bool placed = false;
        for (int probe = 0; probe < 8192; ++probe) {
            // a slot only ever goes from vacant to a key: a plain load that shows this hash (or another one) is final,
            // one that shows a vacant slot is settled by the CAS
            unsigned long long cur = keys[slot];
            if (cur == kEmptyKey) cur = atomicCAS(&keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)h);
            if (cur == kEmptyKey || cur == h) {
                atomicAdd(&a.cnts[slot], c);
                placed = true;
                break;
            }
            slot = (slot + 1) & a.slot_mask;
        }
        if (!placed) atomicOr(reinterpret_cast<unsigned long long *>(a.stats) + kStatFlags, (unsigned long long)kFlagTableFull);
    }
}

hipError_t launch_slab_insert(const SlabMergeArgs &a, uint64_t max_n, hipStream_t st)
{
    uint64_t blocks = (max_n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(slab_insert_kernel, dim3((unsigned)blocks, a.nranks), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// finish(), large sketches: the result block in hash order, written straight into the pinned host block.  The
// extracted hashes are close to uniform below the threshold T (block word [1]), so a counting sort on their leading
// log2(nbuckets) bits below T's top bit puts all but a few neighbours in place: the extract kernel counts the buckets as
// it appends (order_cursor), order_scan_kernel turns counts into start positions (per group of 1024 buckets; the group
// totals are summed by whoever needs them), order_scatter_kernel places the
// entries, order_place_kernel -- one thread per bucket -- ranks the handful of entries of its bucket, stores them at
// their final position in host memory (neighbouring threads, neighbouring addresses) and clears its counter for the
// next finish().  Nothing is read back in between and no copy follows; the host checks the order of what it received
// (buckets beyond kOrderMaxBucket entries are passed through unranked) and sorts itself if it has to.
// Block layout as written by table_extract_kernel: [0] n  [1] T  [2] flags  [3] max-key count  [4 .. 4+cap) hashes
// [4+cap ..) counts.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kOrderMaxBucket = 48;
constexpr uint32_t kScanChunk = 1024;    // counters per workgroup of the scan (256 threads x one uint4)
constexpr uint32_t kMaxScanGroups = 1024; // -> at most 2^20 buckets

// exclusive prefix sum over the 256 threads of a workgroup (value per thread) -> prefix, and the total for everybody
__device__ __forceinline__ uint32_t scan256(uint32_t value, uint32_t *wave_sums /*[4] LDS*/, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v = value;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    if (lane == 63) wave_sums[wave] = v;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t x = wave_sums[w];
        if (w < wave) base += x;
        all += x;
    }
    total = all;
    return base + v - value;
}

// counts -> start positions WITHIN each group of kScanChunk buckets (coalesced, one uint4 per thread), group totals aside;
// the consumers add the groups in front themselves (group_bases)
__global__ __launch_bounds__(256) void order_scan_kernel(uint32_t *cursor, uint32_t *starts, uint32_t *group_total)
{
    __shared__ uint32_t wave_sums[4];
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint4 c = reinterpret_cast<const uint4 *>(cursor)[v];
    uint32_t total;
    uint32_t run = scan256(c.x + c.y + c.z + c.w, wave_sums, total);
    uint4 o;
    o.x = run; run += c.x;
    o.y = run; run += c.y;
    o.z = run; run += c.z;
    o.w = run;
    reinterpret_cast<uint4 *>(cursor)[v] = o;
    reinterpret_cast<uint4 *>(starts)[v] = o;
    if (threadIdx.x == 0) group_total[blockIdx.x] = total;
}

// sbase[g] = entries in the groups in front of group g, sbase[ngroups] = all of them (256 threads, ngroups <= 1024)
__device__ __forceinline__ void group_bases(const uint32_t *group_total, uint32_t ngroups, uint32_t *sbase, uint32_t *wave_sums)
{
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t g = 4u * threadIdx.x + j;
        c[j] = g < ngroups ? group_total[g] : 0u;
        sum += c[j];
    }
    uint32_t total;
    uint32_t run = scan256(sum, wave_sums, total);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t g = 4u * threadIdx.x + j;
        if (g < ngroups) sbase[g] = run;
        run += c[j];
    }
    if (threadIdx.x == 0) sbase[ngroups] = total;
    __syncthreads();
}

__global__ __launch_bounds__(256) void order_scatter_kernel(const uint64_t *blk, uint32_t cap, uint32_t log2_buckets, uint32_t *cursor,
                                                            const uint32_t *group_total, uint64_t *out)
{
    __shared__ uint32_t sbase[kMaxScanGroups + 1], wave_sums[4];
    group_bases(group_total, (1u << log2_buckets) / kScanChunk, sbase, wave_sums);
    const uint32_t n = blk[0] < cap ? (uint32_t)blk[0] : cap;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = blk[4 + i];
    const uint32_t b = (uint32_t)(key >> order_shift(blk[1], log2_buckets));
    const uint32_t d = sbase[b / kScanChunk] + atomicAdd(&cursor[b], 1u);
    if (d >= cap) return; // cannot happen with counters that start from zero
    out[4 + d] = key;
    reinterpret_cast<uint32_t *>(out + 4 + cap)[d] = reinterpret_cast<const uint32_t *>(blk + 4 + cap)[i];
}

__global__ __launch_bounds__(256) void order_place_kernel(const uint64_t *blk, const uint64_t *grouped, uint32_t cap, uint32_t nbuckets,
                                                          uint32_t *cursor, const uint32_t *starts, const uint32_t *group_total, uint64_t *host_blk)
{
    __shared__ uint32_t sbase[kMaxScanGroups + 1], wave_sums[4];
    group_bases(group_total, nbuckets / kScanChunk, sbase, wave_sums);
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < 4) host_blk[b] = blk[b];
    if (b >= nbuckets) return;
    cursor[b] = 0;
    const uint32_t n = blk[0] < cap ? (uint32_t)blk[0] : cap;
    uint32_t lo = sbase[b / kScanChunk] + starts[b], hi = sbase[(b + 1) / kScanChunk] + starts[b + 1]; // starts[nbuckets] = 0 for good
    lo = lo < n ? lo : n;
    hi = hi < n ? hi : n;
    const uint32_t *gc = reinterpret_cast<const uint32_t *>(grouped + 4 + cap);
    uint32_t *hc = reinterpret_cast<uint32_t *>(host_blk + 4 + cap);
    const bool rank_them = hi - lo <= kOrderMaxBucket;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint64_t key = grouped[4 + i];
        uint32_t rank = i - lo;
        if (rank_them && hi - lo > 1) {
            rank = 0;
            for (uint32_t j = lo; j < hi; ++j) rank += grouped[4 + j] < key ? 1u : 0u; // hashes of one table are distinct
        }
        host_blk[4 + lo + rank] = key;
        hc[lo + rank] = gc[i];
    }
}

hipError_t launch_order_block(const uint64_t *blk, uint32_t cap, uint32_t log2_buckets, uint32_t *cursor, uint32_t *starts,
                              uint32_t *group_total, uint64_t *grouped, uint64_t *host_blk, hipStream_t st)
{
    const uint32_t nb = 1u << log2_buckets; // 2^10 .. 2^20
    if (nb < kScanChunk || nb / kScanChunk > kMaxScanGroups) return hipErrorInvalidValue;
    const unsigned blocks = (cap + 255) / 256;
    hipLaunchKernelGGL(order_scan_kernel, dim3(nb / kScanChunk), dim3(256), 0, st, cursor, starts, group_total);
    hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(256), 0, st, blk, cap, log2_buckets, cursor, group_total, grouped);
    hipLaunchKernelGGL(order_place_kernel, dim3(nb / 256), dim3(256), 0, st, blk, grouped, cap, nb, cursor, starts, group_total, host_blk);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// mash compareSketches for one (query, ref) pair per workgroup (Mash 2.x
// CommandDistance.cpp; invoked by /root/reference/auriclass/classes.py:92-104).
// The two ascending lists are merged along 256 merge-path diagonals.  In merged order
// (ties: ref copy first) the query copy of a shared hash directly follows the ref copy, so
//   common = #query copies whose distinct-rank (position - shared copies so far) <= s
//   denom  = min(s, |ref| + |qry| - shared)
// which is exactly what the sequential two-pointer loop with its tail completion yields.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t merge_path(const uint64_t *A, uint32_t nA, const uint64_t *B, uint32_t nB, uint32_t diag)
{ // number of A elements among the first `diag` merged elements (A first on ties)
    uint32_t lo = diag > nB ? diag - nB : 0, hi = diag < nA ? diag : nA;
    while (lo < hi) {
        const uint32_t i = (lo + hi) >> 1, j = diag - 1 - i;
        if (A[i] <= B[j]) lo = i + 1; else hi = i;
    }
    return lo;
}

__global__ __launch_bounds__(256) void dist_pairs_kernel(const DistArgs a)
{
    __shared__ uint32_t part[256];
    __shared__ uint32_t total_common;
    const uint32_t pair = blockIdx.x;
    const uint32_t qi = pair / a.nr, ri = pair % a.nr;
    const uint64_t *A = a.r + (uint64_t)ri * a.stride; // ref
    const uint64_t *B = a.q + (uint64_t)qi * a.stride; // query
    const uint32_t nA = a.r_len[ri], nB = a.q_len[qi];
    const uint32_t total = nA + nB;
    const uint32_t t = threadIdx.x;
    const uint32_t per = (total + 255) / 256;
    const uint32_t d0 = min(t * per, total), d1 = min(d0 + per, total);
    const uint32_t i0 = merge_path(A, nA, B, nB, d0), i1 = merge_path(A, nA, B, nB, d1);
    const uint32_t j0 = d0 - i0, j1 = d1 - i1;
    // pass 1: shared copies in my segment (a query element equal to the ref element before it)
    uint32_t c = 0;
    {
        uint32_t i = i0, j = j0;
        while (i < i1 || j < j1) {
            if (i < i1 && (j >= j1 || A[i] <= B[j])) ++i;
            else { if (i > 0 && A[i - 1] == B[j]) ++c; ++j; }
        }
    }
    part[t] = c;
    if (t == 0) total_common = 0;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t x = 0; x < 256; ++x) { const uint32_t v = part[x]; all += v; if (x < t) before += v; }
    // pass 2: count the shared copies whose distinct rank is within s
    uint32_t counted = 0;
    if (c) {
        uint32_t i = i0, j = j0, pos = d0, seen = before;
        while (i < i1 || j < j1) {
            ++pos;
            if (i < i1 && (j >= j1 || A[i] <= B[j])) ++i;
            else {
                if (i > 0 && A[i - 1] == B[j]) { ++seen; if (pos - seen <= a.s) ++counted; }
                ++j;
            }
        }
    }
    if (counted) atomicAdd(&total_common, counted);
    __syncthreads();
    if (t == 0) {
        const uint32_t uni = total - all;
        const uint32_t denom = uni < a.s ? uni : a.s;
        const uint32_t common = total_common;
        const uint64_t out = (uint64_t)qi * a.out_stride + a.out_off + ri; // the references may be a slice of a wider batch
        a.common[out] = common;
        a.denom[out] = denom;
        if (a.dist) {
            double d;
            if (common == denom) d = 0.0;
            else if (common == 0) d = 1.0;
            else {
                const double jac = (double)common / (double)denom;
                d = -log(2.0 * jac / (1.0 + jac)) / (double)a.k;
                if (d > 1.0) d = 1.0;
            }
            a.dist[out] = d;
        }
    }
}

hipError_t launch_dist_pairs(const DistArgs &a, hipStream_t st)
{
    const uint64_t pairs = (uint64_t)a.nq * a.nr;
    if (pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(dist_pairs_kernel, dim3((unsigned)pairs), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// All-vs-refs distance, fast path.  Hash values are uniform, so cutting the value space into
// 512 equal ranges cuts every sorted list into 512 short, aligned sub-lists (offsets by
// binary search).  For one range, ALL references' hashes (<= 32 refs) go into one LDS hash
// table: key -> bit mask of the references that contain it.  Every query element is then
// probed ONCE and yields its shared-hash bits for all references at the same time (ballot +
// popcount per reference), instead of being merged 24 times.  A last kernel walks the
// per-range counts of each (query, ref) pair to the range where the union reaches s and
// finishes that one short range exactly with the sequential two-pointer rule.
// ---------------------------------------------------------------------------------------
__global__ void dist_shift_kernel(const DistArgs a, DistWork w)
{
    __shared__ unsigned long long gmax;
    if (threadIdx.x == 0) gmax = 0;
    __syncthreads();
    unsigned long long m = 0;
    for (uint32_t i = threadIdx.x; i < a.nq + a.nr; i += blockDim.x) {
        const bool isq = i < a.nq;
        const uint32_t li = isq ? i : i - a.nq;
        const uint32_t n = isq ? a.q_len[li] : a.r_len[li];
        if (n) {
            const uint64_t v = (isq ? a.q : a.r)[(uint64_t)li * a.stride + n - 1];
            m = v > m ? v : m;
        }
    }
    atomicMax(&gmax, m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int bits = gmax ? 64 - __builtin_clzll(gmax) : 1;
        const int lg = 31 - __builtin_clz((unsigned)kDistRanges);
        w.params[0] = bits > lg ? bits - lg : 0; // value >> shift is a range index < kDistRanges
        w.params[1] = 0;
    }
}

// offs[list][p] = first element of the list whose range index (value >> shift) is >= p, for p = 0 .. kDistRanges.
// One thread per ELEMENT: it compares its range index with its left neighbour's and writes the few offsets that fall
// between the two (none at all for 98 % of the elements: a range holds ~49 of them).  Every list is read once, coalesced --
// the per-offset binary searches of round 1 moved 709 MB per C5 call for 419 MB of lists and ran at the HBM limit.
__global__ __launch_bounds__(256) void dist_split_kernel(const DistArgs a, DistWork w, uint32_t list0)
{
    const uint32_t list = blockIdx.x + list0, per = kDistRanges + 1; // lists along x (no 65 535 limit), element blocks along y
    const bool isq = list < a.nq;
    const uint32_t li = isq ? list : list - a.nq;
    const uint32_t n = isq ? a.q_len[li] : a.r_len[li];
    const uint64_t *v = (isq ? a.q : a.r) + (uint64_t)li * a.stride;
    uint32_t *offs = (isq ? w.offs_q : w.offs_r) + li * per;
    const uint32_t shift = w.params[0];
    // two elements per thread: one 16-byte load where the row is 16-byte aligned (8-byte loads run at 0.55-0.7x the rate)
    const uint32_t i = 2 * (blockIdx.y * blockDim.x + threadIdx.x);
    if (n == 0) {
        if (blockIdx.y == 0) for (uint32_t p = threadIdx.x; p < per; p += blockDim.x) offs[p] = 0;
        return;
    }
    if (i >= n) return;
    uint64_t e0, e1 = 0;
    const bool two = i + 1 < n;
    if (two && (reinterpret_cast<uintptr_t>(v + i) & 15) == 0) {
        const uint4 q = *reinterpret_cast<const uint4 *>(v + i);
        e0 = ((uint64_t)q.y << 32) | q.x;
        e1 = ((uint64_t)q.w << 32) | q.z;
    } else {
        e0 = v[i];
        if (two) e1 = v[i + 1];
    }
    const uint32_t r0 = (uint32_t)(e0 >> shift), r1 = two ? (uint32_t)(e1 >> shift) : r0;
    // offsets p in (range of the left neighbour, range of this element] point at this element; the list's first element
    // also serves p = 0 .. its own range, the last one leaves everything above its range at n
    uint32_t from = i == 0 ? 0u : (uint32_t)(v[i - 1] >> shift) + 1u;
    for (uint32_t p = from; p <= r0 && p < per; ++p) offs[p] = i;
    if (two) for (uint32_t p = r0 + 1; p <= r1 && p < per; ++p) offs[p] = i + 1;
    if (i + 2 >= n)
        for (uint32_t p = r1 + 1; p < per; ++p) offs[p] = n;
}

// Sum over the wave of a word of four byte counters (no carry between the bytes as long as every total stays < 256),
// by DPP row operations; lane 63 ends up with the totals.
__device__ __forceinline__ uint32_t wave_sum_bytes(uint32_t v)
{
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false); // row_shr:8  -> lane 15 of every row holds the row's sum
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}

// One reference hash into the range's LDS table: key -> bit mask of the references that hold it.  `ndistinct` counts the
// keys (slots claimed).  Whether a range fits is decided by THAT number, not by the sum of the references' slice sizes:
// references of one clade share most of their hashes (AuriClass's 24 C. auris references do), so 24 slices of 65 hashes
// are 70 keys, not 1560 -- with the sum as the test every such reference set fell back to the generic kernel (0.48 ms
// instead of 0.05 for AuriClass's own 1 x 24 comparison, 62 ms instead of 0.4 for 1024 queries).  The probe sequence is
// bounded by the table size, so a table that does fill up (non-uniform values) ends the build instead of hanging it; the
// caller checks `ndistinct` against kDistTableLimit behind the barrier and gives the range up.
constexpr uint32_t kDistTableLimit = (kDistTableSlots * 3) / 4;
// returns the number of keys this call added (0 or 1; kDistTableSlots when the table had no room at all): the callers sum
// it per thread and add the wave totals to the shared count once, behind the build (dist_table_count)
__device__ __forceinline__ uint32_t dist_table_insert(unsigned long long *keys, uint32_t *masks, uint64_t v, uint32_t r)
{
    uint32_t sl = (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> 40) & (kDistTableSlots - 1);
#pragma nounroll
    for (int probe = 0; probe < kDistTableSlots; ++probe) {
        const unsigned long long prev = atomicCAS(&keys[sl], (unsigned long long)kEmptyKey, (unsigned long long)v);
        if (prev == kEmptyKey || prev == v) { atomicOr(&masks[sl], 1u << r); return prev == kEmptyKey ? 1u : 0u; }
        sl = (sl + 1) & (kDistTableSlots - 1);
    }
    return (uint32_t)kDistTableSlots;
}
__device__ __forceinline__ void dist_table_count(uint32_t *ndistinct, uint32_t mine)
{ // all lanes of the wave call this
#pragma unroll
    for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(ndistinct, mine);
}

__global__ __launch_bounds__(256) void dist_range_kernel(const DistArgs a, DistWork w)
{
    __shared__ unsigned long long keys[kDistTableSlots];
    __shared__ uint32_t masks[kDistTableSlots];
    __shared__ uint32_t too_big;
    // neighbouring ranges share the cache lines their slices begin and end in: consecutive workgroups go round the eight
    // XCDs, so this order puts ranges p, p + 1, ... of one eighth of the value space on ONE XCD (its L2), close in time
    const uint32_t p = (blockIdx.x & 7u) * (kDistRanges / 8) + (blockIdx.x >> 3), per = kDistRanges + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kDistTableSlots; i += 256) { keys[i] = kEmptyKey; masks[i] = 0; }
    if (tid == 0) too_big = 0; // number of distinct keys in the table
    __syncthreads();
    auto slot_of = [](uint64_t x) { return (uint32_t)((x * 0x9E3779B97F4A7C15ull) >> 40) & (kDistTableSlots - 1); };
    // build: wave w inserts references w, w+4, ...; a reference's slice of this range is a
    // few dozen hashes, so all slices of the wave are loaded first, then inserted
    {
        constexpr int G = 8; // 32 references / 4 waves
        uint64_t x[G];
        bool have[G];
        uint32_t added = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t r = wave + 4 * g;
            have[g] = false;
            if (r < a.nr) {
                const uint32_t b = w.offs_r[r * per + p], e = w.offs_r[r * per + p + 1];
                if (b + lane < e) { x[g] = a.r[(uint64_t)r * a.stride + b + lane]; have[g] = true; }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t r = wave + 4 * g;
            if (r >= a.nr) continue;
            const uint32_t b = w.offs_r[r * per + p], e = w.offs_r[r * per + p + 1];
            for (uint32_t i = b + lane; i < e; i += 64) {
                const uint64_t v = (i == b + lane && have[g]) ? x[g] : a.r[(uint64_t)r * a.stride + i];
                added += dist_table_insert(keys, masks, v, r);
            }
        }
        dist_table_count(&too_big, added);
    }
    __syncthreads();
    if (too_big > kDistTableLimit) { // non-uniform input: the host reruns the generic kernel (uniform exit: the count is shared)
        if (tid == 0) atomicOr(&w.params[1], 1u);
        return;
    }
    // probe: wave w takes queries q0+w, q0+w+4, ... of this block's chunk, 8 at a time so that
    // eight global loads are in flight per lane; each query element is looked up once and its reference mask (one bit
    // per reference) is spread into byte counters, four references to a word; one DPP reduction per word and query
    // slice leaves the shared-hash counts of all references in lane 63, which stores them as bytes.
    const uint32_t qper = (a.nq + gridDim.y - 1) / gridDim.y;
    const uint32_t q0 = blockIdx.y * qper, q1 = min(a.nq, q0 + qper);
    const uint32_t nwords = (a.nr + 3) / 4;
    constexpr int G = 8;
    for (uint32_t qb = q0 + wave; qb < q1; qb += 4 * G) {
        uint64_t x[G];
        uint32_t bb[G], ee[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t q = qb + 4 * g;
            bb[g] = ee[g] = 0;
            if (q < q1) { bb[g] = w.offs_q[q * per + p]; ee[g] = w.offs_q[q * per + p + 1]; }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            x[g] = kEmptyKey;
            if (bb[g] + lane < ee[g]) x[g] = a.q[(uint64_t)(qb + 4 * g) * a.stride + bb[g] + lane];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t q = qb + 4 * g;
            if (q >= q1) break;
            if (ee[g] - bb[g] > 255u) { // a byte counter could overflow: not a uniform input, the generic kernel takes over
                if (lane == 0) atomicOr(&w.params[1], 1u);
                continue;
            }
            uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t i0 = bb[g]; i0 < ee[g]; i0 += 64) {
                uint32_t m = 0;
                if (i0 + lane < ee[g]) {
                    const uint64_t v = i0 == bb[g] ? x[g] : a.q[(uint64_t)q * a.stride + i0 + lane];
                    uint32_t sl = slot_of(v);
                    for (;;) {
                        const unsigned long long kx = keys[sl];
                        if (kx == v) { m = masks[sl]; break; }
                        if (kx == kEmptyKey) break;
                        sl = (sl + 1) & (kDistTableSlots - 1);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) // bits 4j .. 4j+3 of the mask -> the low bit of four bytes
                    if (j < (int)nwords) acc[j] += (((m >> (4 * j)) & 0xFu) * 0x00204081u) & 0x01010101u;
            }
            uint8_t *dst = w.cpart + ((uint64_t)q * kDistRanges + p) * (4 * nwords);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j >= (int)nwords) break;
                const uint32_t tot = wave_sum_bytes(acc[j]);
                if (lane == 63) reinterpret_cast<uint32_t *>(dst)[j] = tot;
            }
        }
    }
}

// The same pass with ONE QUERY PER LANE (round 3; used when a chunk holds enough queries to fill the lanes): lane t walks
// the slice of query q0 + t in this range element by element -- load, table probe, spread of the reference mask into
// byte counters -- so that nothing has to be reduced across lanes: the kernel above spends 60 of its ~170 VALU
// instructions per (query, range) slice on the wave-wide tally of a slice that fills 49 of 64 lanes once, this one spends
// ~40 per ELEMENT ROUND of 64 slices, i.e. a quarter of the instructions per element.  Each lane reads its own row
// (16-byte loads where the pair is aligned: a 64-byte line serves four loads of the same lane out of L1/L2), the rows
// of a workgroup's 256 queries are 256 concurrent streams.
#ifndef MHX_DIST_LANE_BLOCK
#define MHX_DIST_LANE_BLOCK 512
#endif
constexpr int kLaneBlock = MHX_DIST_LANE_BLOCK; // queries (= threads) per workgroup: they share one table build
__global__ __launch_bounds__(kLaneBlock) void dist_range_lane_kernel(const DistArgs a, DistWork w)
{
    __shared__ unsigned long long keys[kDistTableSlots];
    __shared__ uint32_t masks[kDistTableSlots];
    __shared__ uint32_t ndistinct; // keys in the table
    const uint32_t p = (blockIdx.x & 7u) * (kDistRanges / 8) + (blockIdx.x >> 3), per = kDistRanges + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // where range p begins and ends in every reference: lane r of EVERY wave holds reference r's pair (nr <= 32), so all
    // reference loads of the build are issued together -- two HBM round trips for the whole build instead of two per
    // reference and wave
    uint32_t rb = 0, re = 0;
    if ((uint32_t)lane < a.nr) { rb = w.offs_r[lane * per + p]; re = w.offs_r[lane * per + p + 1]; }
    for (int i = tid; i < kDistTableSlots; i += kLaneBlock) { keys[i] = kEmptyKey; masks[i] = 0; }
    if (tid == 0) ndistinct = 0;
    auto slot_of = [](uint64_t x) { return (uint32_t)((x * 0x9E3779B97F4A7C15ull) >> 40) & (kDistTableSlots - 1); };
    uint32_t added = 0;
    auto insert = [&](uint64_t v, uint32_t r) { added += dist_table_insert(keys, masks, v, r); };
    constexpr int kWaves = kLaneBlock / 64, kPerWave = (32 + kWaves - 1) / kWaves;
    uint64_t rv[kPerWave];
    uint32_t have = 0;
#pragma unroll
    for (int t = 0; t < kPerWave; ++t) { // build: wave w inserts references w, w + #waves, ...; their first 64 elements
        const uint32_t r = (uint32_t)wave + (uint32_t)kWaves * t;
        const uint32_t b = __shfl(rb, (int)(r & 31u)), e = __shfl(re, (int)(r & 31u));
        rv[t] = 0;
        if (r < a.nr && b + lane < e) { rv[t] = a.r[(uint64_t)r * a.stride + b + lane]; have |= 1u << t; }
    }
    __syncthreads(); // the table is clear
#pragma unroll
    for (int t = 0; t < kPerWave; ++t)
        if ((have >> t) & 1u) insert(rv[t], (uint32_t)wave + (uint32_t)kWaves * t);
    for (uint32_t r = wave; r < a.nr; r += kWaves) { // slices of more than 64 elements (rare with uniform hashes)
        const uint32_t b = __shfl(rb, (int)r), e = __shfl(re, (int)r);
        for (uint32_t i = b + 64u + lane; i < e; i += 64) insert(a.r[(uint64_t)r * a.stride + i], r);
    }
    dist_table_count(&ndistinct, added);
    __syncthreads();
    if (ndistinct > kDistTableLimit) { // non-uniform input: the host reruns the generic kernel (uniform exit: the count is shared)
        if (tid == 0) atomicOr(&w.params[1], 1u);
        return;
    }
    const uint32_t q = blockIdx.y * kLaneBlock + tid;
    if (q >= a.nq) return;
    const uint32_t nwords = (a.nr + 3) / 4;
    const uint32_t b = w.offs_q[q * per + p], e = w.offs_q[q * per + p + 1];
    uint32_t *dst = reinterpret_cast<uint32_t *>(w.cpart + ((uint64_t)q * kDistRanges + p) * (4 * nwords));
    if (e - b > 255u) { // a byte counter could overflow: not a uniform input, the generic kernel takes over
        atomicOr(&w.params[1], 1u);
        return;
    }
    const uint64_t *row = a.q + (uint64_t)q * a.stride;
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto one = [&](uint64_t v) {
        uint32_t sl = slot_of(v), m = 0;
        for (;;) {
            const unsigned long long kx = keys[sl];
            if (kx == v) { m = masks[sl]; break; }
            if (kx == kEmptyKey) break;
            sl = (sl + 1) & (kDistTableSlots - 1);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) // bits 4j .. 4j+3 of the mask -> the low bit of four bytes
            if (j < (int)nwords) acc[j] += (((m >> (4 * j)) & 0xFu) * 0x00204081u) & 0x01010101u;
    };
#ifndef MHX_DIST_LINE64
    if ((reinterpret_cast<uintptr_t>(row) & 127) == 0 && (a.stride & 15u) == 0) {
        // whole 128-byte L2 lines, both halves consumed at once.  With one 64-byte half per step (the form below, round 3's
        // first) the lane kernel fetched exactly TWICE the rows' bytes: 65 k lanes per XCD each keep a line and the next in
        // flight, 8 MB against 4 MB of L2, so the other half of a 128-byte L2 line was gone again before its lane came
        // back for it.  C5: 830 -> 564 MB fetched by this kernel, 0.388 -> 0.399 ms (16 instead of 8 element slots per
        // step, more of them masked at the ends of a slice; -DMHX_DIST_LINE64 brings the old form back)
        for (uint32_t i0 = b & ~15u; i0 < e; i0 += 16) {
            uint4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const uint4 *>(row + i0 + 2 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + 2 * u;
                if (i >= b && i < e) one(((uint64_t)x[u].y << 32) | x[u].x);
                if (i + 1 >= b && i + 1 < e) one(((uint64_t)x[u].w << 32) | x[u].z);
            }
        }
    } else
#endif
    if ((reinterpret_cast<uintptr_t>(row) & 63) == 0 && (a.stride & 7u) == 0) { // (rows of whole lines: nothing is read beyond a row)
        // whole 64-byte lines, each fetched ONCE by the one lane that needs it (four 16-byte loads issued together; with
        // a load per pair of elements a line was fetched up to four times, and 1500 concurrent streams per CU do not fit
        // in its L1); the elements of the first and last line that lie outside the slice are skipped
        uint4 nx[4]; // the line after the one being worked on is already on its way
        const uint32_t first = b & ~7u;
        if (first < e) {
#pragma unroll
            for (int u = 0; u < 4; ++u) nx[u] = *reinterpret_cast<const uint4 *>(row + first + 2 * u);
        }
        for (uint32_t i0 = first; i0 < e; i0 += 8) {
            uint4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = nx[u];
            if (i0 + 8 < e) {
#pragma unroll
                for (int u = 0; u < 4; ++u) nx[u] = *reinterpret_cast<const uint4 *>(row + i0 + 8 + 2 * u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 2 * u;
                if (i >= b && i < e) one(((uint64_t)x[u].y << 32) | x[u].x);
                if (i + 1 >= b && i + 1 < e) one(((uint64_t)x[u].w << 32) | x[u].z);
            }
        }
    } else {
        uint32_t i = b;
        if (i < e && ((reinterpret_cast<uintptr_t>(row + i) & 15) != 0)) { one(row[i]); ++i; }
        for (; i + 2 <= e; i += 2) {
            const uint4 x = *reinterpret_cast<const uint4 *>(row + i);
            one(((uint64_t)x.y << 32) | x.x);
            one(((uint64_t)x.w << 32) | x.z);
        }
        if (i < e) one(row[i]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < (int)nwords) dst[j] = acc[j];
}

// ---- the same all-vs-refs pass without a split pass over the queries (round 3) ------------------------------------------
// dist_split_kernel reads every list once more just to find where the 1024 ranges begin: as much HBM traffic as the
// comparison itself.  Here a workgroup takes kDistWalk CONSECUTIVE ranges for its 256 queries, one query per lane: a lane
// only needs to know where its FIRST range begins (dist_segstart_kernel: one binary search per kDistWalk ranges), walks
// on from there -- the next range starts where the value's range index changes -- and leaves the range starts behind for
// the finish kernel (offs_q).  The line a range ends in is still in the lane's registers when the next range begins.
#ifndef MHX_DIST_WALK
#define MHX_DIST_WALK 4
#endif
constexpr uint32_t kDistWalk = MHX_DIST_WALK;

__global__ __launch_bounds__(256) void dist_segstart_kernel(const DistArgs a, DistWork w)
{
    constexpr uint32_t G = kDistRanges / kDistWalk, per = kDistRanges + 1;
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= a.nq * (G + 1)) return;
    const uint32_t q = id / (G + 1), g = id % (G + 1);
    const uint32_t n = a.q_len[q], shift = w.params[0];
    const uint64_t *v = a.q + (uint64_t)q * a.stride;
    uint32_t lo = 0, hi = n; // first index whose range index (value >> shift) is >= g * kDistWalk
    if (g == G) lo = n;
    else if (g != 0) {
        const uint64_t want = (uint64_t)g * kDistWalk;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((v[mid] >> shift) < want) lo = mid + 1; else hi = mid;
        }
    }
    w.offs_q[q * per + g * kDistWalk] = lo;
}

__global__ __launch_bounds__(256) void dist_walk_kernel(const DistArgs a, DistWork w)
{
    __shared__ unsigned long long keys[kDistTableSlots];
    __shared__ uint32_t masks[kDistTableSlots];
    __shared__ uint32_t too_big;
    constexpr uint32_t G = kDistRanges / kDistWalk, per = kDistRanges + 1;
    // consecutive workgroups go round the eight XCDs: neighbouring segments of one eighth of the value space on one XCD
    const uint32_t g = (blockIdx.x & 7u) * (G / 8) + (blockIdx.x >> 3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t shift = w.params[0];
    const uint32_t q = blockIdx.y * 256 + tid;
    const bool live = q < a.nq;
    const uint32_t nwords = (a.nr + 3) / 4;
    const uint32_t n = live ? a.q_len[q] : 0u;
    const uint64_t *row = a.q + (uint64_t)(live ? q : 0u) * a.stride;
    uint32_t i = live ? w.offs_q[q * per + g * kDistWalk] : 0u;
#ifndef MHX_DIST_LINE64
    constexpr uint32_t kLineElems = 16; // a whole 128-byte L2 line per step: both halves are used before it is evicted
#else
    constexpr uint32_t kLineElems = 8;
#endif
    uint4 x[kLineElems / 2] = {};
    uint32_t loaded = 0xFFFFFFFFu; // first element of the line held in x
    auto slot_of = [](uint64_t v) { return (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> 40) & (kDistTableSlots - 1); };
#pragma nounroll
    for (uint32_t r = 0; r < kDistWalk; ++r) {
        const uint32_t p = g * kDistWalk + r;
        for (int s2 = tid; s2 < kDistTableSlots; s2 += 256) { keys[s2] = kEmptyKey; masks[s2] = 0; }
        if (tid == 0) too_big = 0; // number of distinct keys in the table
        __syncthreads();
        uint32_t added = 0;
        for (uint32_t rr = wave; rr < a.nr; rr += 4) { // build: wave w inserts references w, w + 4, ...
            const uint32_t b = w.offs_r[rr * per + p], e = w.offs_r[rr * per + p + 1];
            for (uint32_t j = b + lane; j < e; j += 64) {
                added += dist_table_insert(keys, masks, a.r[(uint64_t)rr * a.stride + j], rr);
            }
        }
        dist_table_count(&too_big, added);
        __syncthreads();
        // non-uniform input (the table overflowed): the host reruns the generic kernel; this range is skipped.  (No return
        // here: an exit between the build and the walk makes hipcc keep 112 instead of 78 VGPRs, four waves per SIMD
        // instead of six, and the walk 15 % slower.)
        if (tid == 0 && too_big > kDistTableLimit) atomicOr(&w.params[1], 1u);
        if (live && too_big <= kDistTableLimit) {
            uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint32_t begin = i;
            bool more = true; // this lane's walk through range p
            while (more) {
                const uint32_t i0 = i & ~(kLineElems - 1u);
                if (i0 >= n) break;
                if (i0 != loaded) {
#pragma unroll
                    for (int u = 0; u < (int)kLineElems / 2; ++u) x[u] = *reinterpret_cast<const uint4 *>(row + i0 + 2 * u);
                    loaded = i0;
                }
#pragma unroll
                for (int u = 0; u < (int)kLineElems; ++u) {
                    const uint32_t idx = i0 + (uint32_t)u;
                    if (!more || idx < i) continue;
                    if (idx >= n) { i = n; more = false; continue; }
                    const uint4 xv = x[u >> 1];
                    const uint64_t v = (u & 1) ? (((uint64_t)xv.w << 32) | xv.z) : (((uint64_t)xv.y << 32) | xv.x);
                    if ((v >> shift) != p) { i = idx; more = false; continue; } // sorted rows: the next range begins here
                    uint32_t sl = slot_of(v), m = 0;
                    for (;;) {
                        const unsigned long long kx = keys[sl];
                        if (kx == v) { m = masks[sl]; break; }
                        if (kx == kEmptyKey) break;
                        sl = (sl + 1) & (kDistTableSlots - 1);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) // bits 4j .. 4j+3 of the mask -> the low bit of four bytes
                        if (j < (int)nwords) acc[j] += (((m >> (4 * j)) & 0xFu) * 0x00204081u) & 0x01010101u;
                }
                if (more) i = i0 + kLineElems;
            }
            if (i > n) i = n;
            if (i - begin > 255u) atomicOr(&w.params[1], 1u); // a byte counter may have overflowed: not a uniform input
            uint32_t *dst = reinterpret_cast<uint32_t *>(w.cpart + ((uint64_t)q * kDistRanges + p) * (4 * nwords));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < (int)nwords) dst[j] = acc[j];
            w.offs_q[q * per + p + 1] = i; // where the next range begins (the finish kernel reads these)
        }
        __syncthreads(); // nobody still probes the table that the next round clears
    }
}

// 16 pairs per workgroup; thread (seg, pair) first sums its 64 ranges, then the 16 threads of
// segment 0 locate the cut segment, walk it range by range and finish the cut range with the
// sequential two-pointer rule.
__global__ __launch_bounds__(256) void dist_finish_kernel(const DistArgs a, DistWork w)
{
    __shared__ uint32_t seg_uni[kDistSegs][16], seg_com[kDistSegs][16];
    constexpr int RPS = kDistRanges / kDistSegs; // ranges per segment
    // the range kernel gave up on this slice (a value range too crowded for its table or for byte counters): cpart holds
    // cells it never wrote, the host discards this result and runs the generic kernel -- nothing to do here, and
    // nothing to be walked on the strength of stale counts
    if (w.params[1] != 0) return;
    const uint32_t pl = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const uint32_t pair = blockIdx.x * 16 + pl;
    const bool live = pair < a.nq * a.nr;
    const uint32_t q = live ? pair / a.nr : 0, r = live ? pair % a.nr : 0, per = kDistRanges + 1;
    const uint32_t *oq = w.offs_q + q * per, *orr = w.offs_r + r * per;
    // shared hashes per range of this pair: bytes, [query][range][4 * ceil(nr / 4)], 24 bytes apart from range to range
    const uint32_t cstride = 4 * ((a.nr + 3) / 4);
    const uint8_t *cp = w.cpart + (uint64_t)q * kDistRanges * cstride + r;
    {
        uint32_t com = 0;
        for (uint32_t p = seg * RPS; p < (seg + 1) * RPS; ++p) com += cp[(uint64_t)p * cstride];
        const uint32_t p0 = seg * RPS, p1 = (seg + 1) * RPS;
        seg_com[seg][pl] = com;
        seg_uni[seg][pl] = (orr[p1] - orr[p0]) + (oq[p1] - oq[p0]) - com;
    }
    __syncthreads();
    if (seg != 0 || !live) return;
    const uint32_t S = a.s;
    uint32_t uni = 0, common = 0, sg = 0;
    for (; sg < kDistSegs; ++sg) {
        if (uni + seg_uni[sg][pl] >= S) break;
        uni += seg_uni[sg][pl];
        common += seg_com[sg][pl];
    }
    uint32_t denom;
    if (sg == kDistSegs) denom = uni; // union smaller than s: everything counts
    else {
        uint32_t p = sg * RPS;
        for (; p + 1 < (sg + 1) * RPS; ++p) { // the cut range is inside this segment (its last range is the cut at the latest)
            const uint32_t c = cp[(uint64_t)p * cstride];
            const uint32_t u = (orr[p + 1] - orr[p]) + (oq[p + 1] - oq[p]) - c;
            if (uni + u >= S) break;
            uni += u;
            common += c;
        }
        const uint64_t *A = a.r + (uint64_t)r * a.stride, *B = a.q + (uint64_t)q * a.stride;
        uint32_t i = orr[p], j = oq[p];
        const uint32_t ie = orr[p + 1], je = oq[p + 1];
        while (uni < S && i < ie && j < je) {
            const uint64_t x = A[i], y = B[j];
            if (x < y) ++i;
            else if (y < x) ++j;
            else { ++i; ++j; ++common; }
            ++uni;
        }
        denom = S; // this range holds enough further union elements by construction
    }
    const uint64_t out = (uint64_t)q * a.out_stride + a.out_off + r; // the references may be a slice of a wider batch
    a.common[out] = common;
    a.denom[out] = denom;
    if (a.dist) {
        double d;
        if (common == denom) d = 0.0;
        else if (common == 0) d = 1.0;
        else {
            const double jac = (double)common / (double)denom;
            d = -log(2.0 * jac / (1.0 + jac)) / (double)a.k;
            if (d > 1.0) d = 1.0;
        }
        a.dist[out] = d;
    }
}

size_t dist_work_bytes(uint32_t nq, uint32_t nr, size_t *off_q, size_t *off_r, size_t *off_c, size_t *off_p)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    *off_q = o; o += up((size_t)nq * (kDistRanges + 1) * 4);
    *off_r = o; o += up((size_t)nr * (kDistRanges + 1) * 4);
    *off_c = o; o += up((size_t)kDistRanges * nq * 4 * ((nr + 3) / 4));
    *off_p = o; // behind it: two words per (query batch, reference slice) block, sized by the caller
    return o;
}

hipError_t launch_dist_ranges(const DistArgs &a, const DistWork &w, hipStream_t st)
{
    hipLaunchKernelGGL(dist_shift_kernel, dim3(1), dim3(256), 0, st, a, w);
    const bool no_lane = getenv("MHX_DIST_NO_LANE") != nullptr, no_walk = getenv("MHX_DIST_NO_WALK") != nullptr;
    // rows of whole 128-byte lines (the walk reads a row line by line), enough queries to fill the lanes.
    // Measured with 128-byte lines in both forms (C5 rows, profiles/r03_dist_line128_ab.txt): 1024 queries 0.40 ms lane
    // form / 0.49 walk (1024 workgroups of four ranges each leave the CUs a third empty), 4096: 1.30 / 1.37, 8192: 2.60 /
    // 2.52 -- the walk takes over there; it fetches 1.9x the algorithmic bytes against the lane form's 2.6x (no split
    // pass over the queries, but 257 binary searches per row).  MHX_DIST_WALK_MIN moves the switch.
    const uint32_t walk_min = getenv("MHX_DIST_WALK_MIN") ? (uint32_t)atol(getenv("MHX_DIST_WALK_MIN")) : 8192u;
    #ifndef MHX_DIST_LINE64
    const bool whole_lines = (a.stride & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.q) & 127) == 0;
#else
    const bool whole_lines = (a.stride & 7u) == 0 && (reinterpret_cast<uintptr_t>(a.q) & 63) == 0;
#endif
    const bool walk = !no_lane && !no_walk && a.nq >= walk_min && whole_lines;
    if (walk) {
        hipLaunchKernelGGL(dist_split_kernel, dim3(a.nr, (a.stride + 511) / 512), dim3(256), 0, st, a, w, a.nq); // the references only
        constexpr uint32_t G = kDistRanges / kDistWalk;
        hipLaunchKernelGGL(dist_segstart_kernel, dim3((a.nq * (G + 1) + 255) / 256), dim3(256), 0, st, a, w);
        hipLaunchKernelGGL(dist_walk_kernel, dim3(G, (a.nq + 255) / 256), dim3(256), 0, st, a, w);
    } else {
        hipLaunchKernelGGL(dist_split_kernel, dim3(a.nq + a.nr, (a.stride + 511) / 512), dim3(256), 0, st, a, w, 0u);
        if (a.nq >= 128 && !no_lane) hipLaunchKernelGGL(dist_range_lane_kernel, dim3(kDistRanges, (a.nq + kLaneBlock - 1) / kLaneBlock), dim3(kLaneBlock), 0, st, a, w);
        else hipLaunchKernelGGL(dist_range_kernel, dim3(kDistRanges, kDistQueryChunks), dim3(256), 0, st, a, w);
    }
    const uint32_t pairs = a.nq * a.nr;
    hipLaunchKernelGGL(dist_finish_kernel, dim3((pairs + 15) / 16), dim3(256), 0, st, a, w);
    return hipGetLastError();
}

} // namespace mhx
