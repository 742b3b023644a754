#!/usr/bin/env python3
"""profiles/traffic.json from a round's PMC summary and kernel trace (what bench.py quotes as roofline.traffic and
valu_issue for the exact workload the counters were collected on: the default bench command).
usage: tools/make_traffic.py <pmc_summary.json> <kernel_stats_trace.csv> <bench_line_under_trace.json> <out.json> <round-tag>"""
import csv
import json
import sys

pmc, trace, line, out, tag = sys.argv[1:6]
j = json.load(open(pmc))
steps = j["steps_per_pmc_run"]
tile = {k: v for k, v in j["kernels"].items() if "sketch_tile_kernel" in k}
rd = sum(v.get("hbm_read_bytes_per_step_corrected", 0.0) for v in tile.values())
wr = sum(v.get("hbm_write_bytes_per_step", 0.0) for v in tile.values())
valu = sum(v.get("SQ_INSTS_VALU", 0.0) for v in tile.values()) / steps
gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in tile.values()) / steps / 8.0   # summed over the 8 XCDs
bl = json.loads(open(line).read().strip().splitlines()[-1])
passes = bl["steps"] + bl["warmup"] + 3   # bench.py: warmup + timed + 3 event-timed passes
tile_ns = 0.0
for r in csv.DictReader(open(trace)):
    if "sketch_tile_kernel" in r["Name"]:
        tile_ns += float(r["TotalDurationNs"])
kernel_s = tile_ns / passes * 1e-9
res = {
    "note": f"HBM bytes per bench step of sketch_tile_kernel (all forms summed) from the separate --pmc passes of tools/profile.sh "
            f"(FETCH_SIZE x 1024 x 2 per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE x 1024), round {tag}",
    "algorithmic_bytes_per_step": bl["roofline"]["algorithmic_bytes_per_step"],
    "sketch_tile_kernel_hbm_read_bytes_per_step": round(rd),
    "sketch_tile_kernel_hbm_write_bytes_per_step": round(wr),
    "sketch_tile_kernel_hbm_bytes_per_step": round(rd + wr),
    "sketch_tile_kernel_valu_wave_instructions_per_step": round(valu),
    "valu_note": "SQ_INSTS_VALU of sketch_tile_kernel per bench step; a wave64 integer VALU instruction occupies its SIMD for ~4 cycles "
                 "(profiles/r02_valu_class_rates_microbench.txt)",
    "shader_clock_ghz_from_pmc": round(gui / kernel_s / 1e9, 3) if kernel_s else None,
    "clock_note": f"GRBM_GUI_ACTIVE per XCD and step / tile-kernel seconds per step under rocprofv3 --kernel-trace ({kernel_s * 1e3:.3f} ms)",
    "kernel_ms_per_step_under_trace": round(kernel_s * 1e3, 4),
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
