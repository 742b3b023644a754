// inflate_fuzz.cpp -- AddressSanitizer/UBSan harness of the ingest's gzip decoder (CPU build only: GPU ASan is
// not available).  Built by tests/test_lib_cpu.py from auriclass_amd/csrc/mhx_inflate.cpp + this file:
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all ...
// Every case gets its input in an exact-size heap block (n + GzInflater::kInputPad bytes), so a load past the
// promised pad is a heap-buffer-overflow report, and its output in an exact-size block behind a 32 KiB history.
// usage: inflate_fuzz <seed.gz> <mutations> <rng seed>     -> exit 0, or the sanitizer aborts
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../auriclass_amd/csrc/mhx_internal.h"

namespace mhx {
int fail(int code, const char *, ...) { return code; }
void clear_error() {}
} // namespace mhx

using mhx::GzInflater;

static uint64_t rng_state = 1;
static uint64_t rnd()
{ // xorshift64*
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return rng_state * 0x2545F4914F6CDD1Dull;
}

// returns bytes produced, or -1 when the decoder refused the stream
static long run_case(const uint8_t *z, size_t n, bool deferred)
{
    uint8_t *in = (uint8_t *)malloc(n + GzInflater::kInputPad);
    memcpy(in, z, n);
    memset(in + n, 0, GzInflater::kInputPad);
    GzInflater inf;
    inf.set_input(in, n);
    inf.set_deferred_crc(deferred);
    const size_t piece = 1u << 16;
    const size_t cap = GzInflater::kWindow + piece + GzInflater::kOvershoot + 16;
    uint8_t *buf = (uint8_t *)malloc(cap);
    size_t hist = 0;
    long total = 0;
    for (int rounds = 0; rounds < 100000; ++rounds) {
        uint8_t *o = buf + GzInflater::kWindow;
        const size_t got = inf.inflate(o, piece, o - hist);
        if (got == (size_t)-1) { total = -1; break; }
        uint32_t crc;
        while (inf.take_member_end(&crc)) { }
        total += (long)got;
        if (inf.done()) break;
        const size_t have = hist + got, keep = have < GzInflater::kWindow ? have : GzInflater::kWindow;
        memmove(buf + GzInflater::kWindow - keep, o + got - keep, keep);
        hist = keep;
    }
    free(buf);
    free(in);
    return total;
}

// the same input through the parallel decoder (mhx_pinflate.cpp); -1 refused, -2 declined (falls to the sequential decoder)
static long run_par_case(const uint8_t *z, size_t n, int threads)
{
    uint8_t *in = (uint8_t *)malloc(n + GzInflater::kInputPad);
    memcpy(in, z, n);
    memset(in + n, 0, GzInflater::kInputPad);
    long total = 0;
    bool was_bgzf = false;
    {
        mhx::BgzfReader bg; // bgzip's blocks first, as the ingest does
        if (bg.start(in, n, threads)) {
            was_bgzf = true;
            std::vector<uint8_t> piece(1u << 16);
            for (;;) {
                const size_t got = bg.read(piece.data(), piece.size());
                if (got == (size_t)-1) { total = -1; break; }
                if (got == 0) break;
                total += (long)got;
            }
        }
    }
    if (!was_bgzf) {
        mhx::ParallelGunzip par;
        if (!par.start(in, n, threads)) total = -2;
        else {
            std::vector<uint8_t> piece(1u << 16);
            for (;;) {
                const size_t got = par.read(piece.data(), piece.size());
                if (got == (size_t)-1) { total = -1; break; }
                if (got == 0) break;
                total += (long)got;
            }
        }
    }
    free(in);
    return total;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> seed;
    uint8_t tmp[65536];
    size_t r;
    while ((r = fread(tmp, 1, sizeof(tmp), f)) > 0) seed.insert(seed.end(), tmp, tmp + r);
    fclose(f);
    const long mutations = atol(argv[2]);
    rng_state = (uint64_t)atoll(argv[3]) * 2 + 1;
    long ok = 0, refused = 0, par_ok = 0, par_refused = 0, par_declined = 0;
    auto tally = [&](long v) { if (v < 0) ++refused; else ++ok; };
    // 1. the seed itself must decode, both ways, to the same length
    const long seed_len = run_case(seed.data(), seed.size(), false);
    if (seed_len < 0) { fprintf(stderr, "seed refused\n"); return 3; }
    const long par_len = run_par_case(seed.data(), seed.size(), 3);
    if (par_len != -2 && par_len > seed_len) { fprintf(stderr, "parallel decoder produced more than the member holds\n"); return 5; }
    // 2. crafted: 10-byte header + a stored block + the start of another stored block whose LEN/NLEN lie in the pad
    {
        const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
        std::vector<uint8_t> z(hdr, hdr + 10);
        const uint8_t blk[] = {0x00, 0x04, 0x00, 0xFB, 0xFF, 'A', 'C', 'G', 'T'};
        z.insert(z.end(), blk, blk + sizeof(blk));
        const uint8_t tails[][3] = {{0x01, 0xFF, 0xFF}, {0x00, 0xFF, 0xFF}, {0x01, 0x00, 0x00}};
        for (auto &t : tails)
            for (int keep = 1; keep <= 3; ++keep) {
                std::vector<uint8_t> y = z;
                y.insert(y.end(), t, t + keep);
                while (y.size() < 18) y.push_back(0);
                if (run_case(y.data(), y.size(), false) >= 0 && keep == 3 && t[1] == 0xFF) { fprintf(stderr, "crafted stored block accepted\n"); return 4; }
            }
    }
    // 3. truncation sweep over the first 600 bytes (member header, first dynamic block header) and the last 40
    for (size_t cut = 0; cut < seed.size(); ++cut) {
        if (cut > 600 && cut + 40 < seed.size()) { cut += 97; if (cut >= seed.size()) break; }
        tally(run_case(seed.data(), cut, (cut & 1) != 0));
    }
    // 4. random mutations: bit flips, byte splats, truncation after a flip
    for (long i = 0; i < mutations; ++i) {
        std::vector<uint8_t> y = seed;
        const int edits = 1 + (int)(rnd() % 4);
        for (int e = 0; e < edits; ++e) {
            const size_t pos = (size_t)(rnd() % y.size());
            switch (rnd() % 3) {
            case 0: y[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
            case 1: y[pos] = (uint8_t)rnd(); break;
            default: { const size_t len = 1 + (size_t)(rnd() % 8); for (size_t j = pos; j < y.size() && j < pos + len; ++j) y[j] = 0xFF; }
            }
        }
        size_t n = y.size();
        if (rnd() % 4 == 0) n = (size_t)(rnd() % (y.size() + 1));
        tally(run_case(y.data(), n, (i & 1) != 0));
        if (i % 4 == 0) { // refused, declined or decoded: no out-of-bounds access either way
            const long pr = run_par_case(y.data(), n, 2 + (int)(i % 3));
            if (pr == -2) ++par_declined; else if (pr == -1) ++par_refused; else ++par_ok;
        }
    }
    printf("ok %ld refused %ld parallel: ok %ld refused %ld declined %ld (seed: %ld)\n", ok, refused, par_ok, par_refused, par_declined, par_len);
    return 0;
}
