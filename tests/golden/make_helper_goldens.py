"""Golden input/output vectors of the reference's small host helpers (general.py, args.py): data only.

Run in the build container only (needs /root/reference, pyfastx stubbed):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_helper_goldens.py
Covers add_tag (general.py:10-32), check_number_within_range (:35-65), validate_argument_logic (:141-177)
and the parsed CLI namespaces of args.py:8-160 for a set of argument vectors.
"""
import argparse
import json
import sys
import types
from pathlib import Path

OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.modules["pyfastx"] = types.ModuleType("pyfastx")
sys.path.insert(0, "/root/reference")
import auriclass.args as ra  # noqa: E402
import auriclass.general as rg  # noqa: E402


def outcome(fn):
    try:
        return {"value": fn()}
    except SystemExit as e:
        return {"raises": "SystemExit", "code": e.code}
    except BaseException as e:
        return {"raises": type(e).__name__}


g = {"add_tag": [], "range": [], "logic": [], "argv": []}
for tag, lines in [("mash sketch", "a\nb\n"), ("x", ""), ("t", "\n\n"), ("mash dist", "one line"), ("q", "a\n\nb")]:
    g["add_tag"].append({"tag": tag, "lines": lines, "out": rg.add_tag(tag, lines)})
for lo, hi in [(0, 1), (1, 32), (5, 1_000_000), (0, 0.4)]:
    for v in ["0", "1", "0.5", "1.0000001", "-1", "32", "33", "abc", "1e3", " 7 ", "0.4", "inf", "nan"]:
        g["range"].append({"min": lo, "max": hi, "value": v, **outcome(lambda: rg.check_number_within_range(lo, hi)(v))})
for pair in [["11400000", "14900000"], ["11.4", "14.9"], ["14.9", "11.4"], ["99", "99"], ["99", "100"], ["100", "99"], ["0", "0"],
             ["12", "14900000"], ["1e7", "2e7"], ["x", "1"]]:
    ns = argparse.Namespace(expected_genome_size=list(pair))
    g["logic"].append({"pair": pair, **outcome(lambda: rg.validate_argument_logic(ns).expected_genome_size)})
vectors = [
    ["reads_1.fq.gz", "reads_2.fq.gz"],
    ["asm.fasta", "-n", "sampleA", "-o", "out.tsv", "--fasta"],
    ["r.fq", "--fastq", "-t", "4", "--no_qc", "--log_file_path", "x.log", "--verbose"],
    ["a", "--kmer_size", "21", "--sketch_size", "1000", "--minimal_kmer_coverage", "2", "--expected_genome_size", "11", "15"],
    ["a", "--kmer_size", "33"], ["a", "--sketch_size", "1"], ["a", "--non_candida_threshold", "2"], ["a", "--fastq", "--fasta"], [],
    ["a", "--high_dist_threshold", "0.5", "--non_candida_threshold", "0.02", "--debug"], ["a", "--version"],
]
for argv in vectors:
    sys.argv = ["auriclass"] + argv

    def parse():
        ns = ra.auriclass_arg_parser()
        return {k: (str(v) if isinstance(v, Path) else v) for k, v in sorted(vars(ns).items())}

    import contextlib
    import io

    with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
        g["argv"].append({"argv": argv, **outcome(parse)})
(OUT / "helper_goldens.json").write_text(json.dumps(g, indent=0, default=str) + "\n")
print({k: len(v) for k, v in g.items()})
print([x.get("raises") for x in g["argv"]])
