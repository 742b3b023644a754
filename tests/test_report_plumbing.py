"""CPU test of the host-side mirror (auriclass_amd.classes / general / args / main): with the
engine's three text-producing calls replaced by the golden mash texts the reference's tests
pin, the QC decisions and the one-row report must be byte-identical to the reference's
tests/data/reference_report_{fastq,fasta}.tsv (K7).  No GPU and no compute involved: this
checks the plumbing around the boundary, not the kernels."""
import argparse

import pytest

from auriclass_amd import classes, engine, general
from auriclass_amd.args import build_parser
from tests.conftest import GOLDEN, REFDATA

FASTQ_DIST = ("tests/data/NC_001416.1.fasta\ttests/data/NC_001416.1_1.fq.gz\t9.55405e-06\t0\t48451/48476\n"
              "tests/data/NC_001604.1.fasta\ttests/data/NC_001416.1_1.fq.gz\t1\t1\t0/50000\n")
FASTA_DIST = ("tests/data/NC_001416.1.fasta\ttests/data/NC_001416.1.fasta.gz\t0\t0\t48476/48476\n"
              "tests/data/NC_001604.1.fasta\ttests/data/NC_001416.1.fasta.gz\t1\t1\t0/50000\n")


@pytest.fixture()
def golden_engine(monkeypatch):
    bounds = (GOLDEN / "mash_bounds_k27_p0.99.txt").read_text()
    state = {"dist": FASTQ_DIST}

    def fake_sketch(paths, k, s, out, reads=False, min_mult=1):
        if any("empty" in str(p) for p in paths):
            raise engine.NoRecordsError(engine.MHX_E_NO_RECORDS, 'ERROR: Did not find fasta records in "x".')
        return ("Estimated genome size: 48454.7\nEstimated coverage:    39.125\n" if reads else "Sketching x...\n"), 48454.7

    monkeypatch.setattr(engine, "sketch_files", fake_sketch)
    monkeypatch.setattr(engine, "dist_files", lambda r, q: state["dist"])
    monkeypatch.setattr(engine, "bounds", lambda k, p: bounds)
    monkeypatch.setattr(engine, "fasta_total_bases", lambda p: 48502)
    return state


def make(cls, refcwd, paths, **kw):
    args = dict(name="isolate", read_paths=paths, output_report_path="tmp_data/report.tsv",
                reference_sketch_path="tests/data/ref_sketch.msh", genome_size_range=[40_000, 60_000], kmer_size=27,
                sketch_size=50_000, minimal_kmer_coverage=3, clade_config_path="tests/data/clade_config.csv",
                non_candida_threshold=0.01, high_dist_threshold=0.003, no_qc=False)
    args.update(kw)
    return cls(**args)


def test_fastq_steps_and_report_bytes(refcwd, golden_engine, golden):
    s = make(classes.FastqAuriclass, refcwd, ["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz"])
    s.query_sketch_path = "tmp_data/q.msh"           # the reference's tests assign a str here
    s.sketch_fastq_query()
    s.run_mash_dist()
    want = golden["mash_output_to_dict_fastq"]
    got = s.mash_output.to_dict()
    assert {c: {str(i): v for i, v in col.items()} for c, col in got.items()} == want
    s.check_genome_size()
    assert s.estimated_genome_size == 48454.7 and s.qc_genome_size == ""
    s.select_clade()
    assert (s.clade, s.minimal_distance, s.closest_sample) == ("Lambda phage", 9.55405e-06, "tests/data/NC_001416.1.fasta")
    assert s.check_non_candida() and s.check_for_outgroup()
    s.check_high_dist()
    s.process_error_bounds(s.get_error_bounds())
    assert s.error_bound == 0.0008979
    s.compare_with_error_bounds()
    assert s.distances == [1.0] and s.samples_within_error_bound == 0 and s.qc_multiple_hits == ""
    s.save_report()
    assert open("tmp_data/report.tsv", "rb").read() == (REFDATA / "reference_report_fastq.tsv").read_bytes()
    # and the whole thing through run()
    s2 = make(classes.FastqAuriclass, refcwd, ["a.fq"], output_report_path="tmp_data/r2.tsv")
    s2.run()
    assert open("tmp_data/r2.tsv", "rb").read() == (REFDATA / "reference_report_fastq.tsv").read_bytes()


def test_fasta_run_report_bytes(refcwd, golden_engine):
    golden_engine["dist"] = FASTA_DIST
    s = make(classes.FastaAuriclass, refcwd, ["tests/data/NC_001416.1.fasta.gz"])
    s.run()
    assert s.estimated_genome_size == 48502 and s.minimal_distance == 0
    assert open("tmp_data/report.tsv", "rb").read() == (REFDATA / "reference_report_fasta.tsv").read_bytes()


def test_decision_branches(refcwd, golden_engine):
    # closest reference is the outgroup -> FAIL, special clade label
    golden_engine["dist"] = FASTQ_DIST.replace("9.55405e-06", "0.5").replace("\t1\t1\t0/50000", "\t0.001\t0\t49000/50000")
    s = make(classes.FastqAuriclass, refcwd, ["a.fq"])
    s.run()
    assert s.clade == "other Candida/CUG-Ser1 clade sp." and s.qc_decision == "FAIL"
    # too far from everything -> not Candida auris, four SKIPPED
    golden_engine["dist"] = FASTQ_DIST.replace("9.55405e-06", "0.2")
    s = make(classes.FastqAuriclass, refcwd, ["a.fq"])
    s.run()
    assert s.clade == "not Candida auris"
    assert [s.qc_other_candida, s.qc_genome_size, s.qc_multiple_hits, s.qc_high_distance] == ["SKIPPED"] * 4
    # --no_qc -> WARN metrics SKIPPED; genome size outside range -> WARN
    golden_engine["dist"] = FASTQ_DIST
    s = make(classes.FastqAuriclass, refcwd, ["a.fq"], no_qc=True, genome_size_range=[1, 2])
    s.run()
    assert s.qc_genome_size == "SKIPPED" and s.qc_decision == "PASS"
    s = make(classes.FastqAuriclass, refcwd, ["a.fq"], genome_size_range=[1, 2], high_dist_threshold=1e-7)
    s.run()
    assert s.qc_decision == "WARN" and s.qc_genome_size.startswith("WARN") and s.qc_high_distance.startswith("WARN")
    # a sketch size without a bounds row raises IndexError, as in the reference
    s = make(classes.FastqAuriclass, refcwd, ["a.fq"], sketch_size=2000)
    with pytest.raises(IndexError):
        s.run()


def test_error_conventions(refcwd, golden_engine):
    # empty input -> ValueError from the sketch step (tests/test_failing_workflow.py:51-78)
    s = make(classes.FastqAuriclass, refcwd, ["tests/data/test_empty_1.fq.gz", "tests/data/test_empty_2.fq.gz"])
    with pytest.raises(ValueError, match="Did not find sequence records"):
        s.sketch_fastq_query()
    with pytest.raises(FileNotFoundError):
        general.validate_input_files(["tests/data/doesnotexist_1.fq.gz"])
    # a sketch given as reads is neither format; mixed inputs are rejected
    with pytest.raises(ValueError, match="not a fastq or fasta"):
        general.guess_input_type(["tests/data/ref_sketch.msh"])
    with pytest.raises(ValueError, match="mix"):
        general.guess_input_type(["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1.fasta.gz"])
    assert general.guess_input_type(["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz"]) == "fastq"
    assert general.guess_input_type(["tests/data/NC_001416.1.fasta.gz"]) == "fasta"


def test_argument_logic_and_parser():
    ns = argparse.Namespace(expected_genome_size=["11.4", "14.9"])
    assert general.validate_argument_logic(ns).expected_genome_size == [11_400_000.0, 14_900_000.0]
    with pytest.raises(ValueError):
        general.validate_argument_logic(argparse.Namespace(expected_genome_size=[5, 1]))
    p = build_parser()
    a = p.parse_args(["x.fq"])
    assert (a.kmer_size, a.sketch_size, a.minimal_kmer_coverage, a.name) == (27, 50_000, 3, "isolate")
    assert a.expected_genome_size == [11_400_000, 14_900_000] and a.reference_sketch_path == ""
    a = p.parse_args(["x.fq", "-k", "21", "-s", "1000", "--non_candida_threshold", "0.5"])
    assert (a.kmer_size, a.sketch_size, a.non_candida_threshold) == ("21", "1000", "0.5")   # validators keep strings
    with pytest.raises(SystemExit):
        p.parse_args(["x.fq", "-k", "40"])
    assert general.add_tag("t", "a\n\nb") == "[t] a\n[t] b" and general.add_tag("t", "") == "[t]"


# ---- 84 scenarios whose expectations were produced by the reference's own Python layer ----------------
def _scenarios():
    import json

    return json.loads((GOLDEN / "report_scenarios.json").read_text())


@pytest.mark.parametrize("sc", _scenarios(), ids=lambda sc: "%s-%d" % (sc["mode"], sc["id"]))
def test_report_scenarios_from_the_reference(tmp_path, monkeypatch, sc):
    """tests/golden/report_scenarios.json: random dist tables over 24 references in 6 clades (clear hits,
    second hits inside the error bound, outgroup hits, ties, nothing close), genome sizes around the
    expected range, thresholds, --no_qc, and the two quirks of the bounds lookup.  The expected report
    rows / exception types come from /root/reference/auriclass/classes.py itself
    (tests/golden/make_report_scenarios.py); here the same texts go through the mirror."""
    bounds = (GOLDEN / "mash_bounds_k27_p0.99.txt").read_text()
    est = sc["estimated_genome_size"]
    monkeypatch.setattr(engine, "sketch_files", lambda paths, k, s, out, reads=False, min_mult=1: (
        ("Estimated genome size: %g\nEstimated coverage:    40\n" % est) if reads else "Sketching x...\n", est))
    monkeypatch.setattr(engine, "dist_files", lambda r, q: sc["dist_text"])
    monkeypatch.setattr(engine, "bounds", lambda k, p: bounds)
    monkeypatch.setattr(engine, "fasta_total_bases", lambda p: int(est))
    clade_csv = tmp_path / "clades.csv"
    clade_csv.write_text("filename,clade\n" + "".join("%s,%s\n" % kv for kv in sc["clades"].items()))
    cls = classes.FastqAuriclass if sc["mode"] == "fastq" else classes.FastaAuriclass
    report = tmp_path / "report.tsv"
    obj = cls(name="isolate", output_report_path=report,
              read_paths=["reads_1.fq.gz", "reads_2.fq.gz"] if sc["mode"] == "fastq" else ["asm.fasta"],
              reference_sketch_path="refs.msh", kmer_size=27, sketch_size=sc["sketch_size"], minimal_kmer_coverage=3,
              clade_config_path=clade_csv, genome_size_range=sc["genome_size_range"],
              non_candida_threshold=sc["non_candida_threshold"], high_dist_threshold=sc["high_dist_threshold"], no_qc=sc["no_qc"])
    want = sc["expect"]
    if "raises" in want:
        with pytest.raises(Exception) as ei:
            obj.run()
        assert type(ei.value).__name__ == want["raises"]
        return
    obj.run()
    assert report.read_text() == want["report"]
    assert obj.clade == want["clade"]
    assert float(obj.minimal_distance) == want["minimal_distance"]
    assert int(obj.samples_within_error_bound) == want["samples_within_error_bound"]
    assert float(obj.error_bound) == want["error_bound"]


# ---- small helpers: vectors produced by the reference's general.py / args.py -----------------------------
def _helper_goldens():
    import json

    return json.loads((GOLDEN / "helper_goldens.json").read_text())


def _outcome(fn):
    try:
        return {"value": fn()}
    except SystemExit as e:
        return {"raises": "SystemExit", "code": e.code}
    except BaseException as e:
        return {"raises": type(e).__name__}


def test_helper_functions_match_the_reference_vectors(capsys):
    """tests/golden/helper_goldens.json (made by tests/golden/make_helper_goldens.py from the reference):
    add_tag, the argparse range validators (they return str, not numbers), the genome-size logic (Mbp below
    100, lower > upper) and the parsed namespace of eleven argument vectors, errors included."""
    from pathlib import Path

    from auriclass_amd.args import auriclass_arg_parser

    g = _helper_goldens()
    for c in g["add_tag"]:
        assert general.add_tag(c["tag"], c["lines"]) == c["out"]
    for c in g["range"]:
        got = _outcome(lambda: general.check_number_within_range(c["min"], c["max"])(c["value"]))
        assert got == {k: v for k, v in c.items() if k in ("value", "raises") and not (k == "value" and "raises" in c)} or \
            got == ({"raises": c["raises"]} if "raises" in c else {"value": c["value"]}), c
    for c in g["logic"]:
        ns = argparse.Namespace(expected_genome_size=list(c["pair"]))
        got = _outcome(lambda: general.validate_argument_logic(ns).expected_genome_size)
        want = {"raises": c["raises"]} if "raises" in c else {"value": c["value"]}
        assert got == want, c
    skip = {"clade_config_path", "reference_sketch_path"}       # defaults depend on where the package is installed
    for c in g["argv"]:
        def parse():
            ns = auriclass_arg_parser(c["argv"])
            return {k: (str(v) if isinstance(v, Path) else v) for k, v in sorted(vars(ns).items()) if k not in skip}
        got = _outcome(parse)
        if "raises" in c:
            assert got.get("raises") == c["raises"] and got.get("code") == c.get("code"), c
        else:
            assert got == {"value": {k: v for k, v in c["value"].items() if k not in skip}}, c
