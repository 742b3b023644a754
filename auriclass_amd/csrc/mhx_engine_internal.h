// mhx_engine_internal.h -- what mhx_engine.cpp (engine state, sketcher, distances) and mhx_files.cpp (file
// ingest and the file-level calls) share.  Internal; the public surface is include/mhx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mhx_internal.h"

namespace mhx {

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(MHX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- engine state ---------------------------------------------------------------------
struct Engine {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_dist_ms = 0.0;
    uint8_t *dist_ws = nullptr; // workspace of the all-vs-refs distance path
    size_t dist_ws_cap = 0;
    // bulk file ingest: pinned staging ring + copy stream (allocated on first use, kept)
    static constexpr int kPinnedSlots = 4;
    uint8_t *pinned[kPinnedSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pinned_free[kPinnedSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
};
extern Engine g;
int require_engine();

} // namespace mhx

// sketcher with `table_scale` times the default candidate table and admission budget
int create_sketcher(int k, uint32_t s, uint32_t min_mult, uint64_t expected_bytes, uint64_t table_scale, mhx_sketcher **out);
