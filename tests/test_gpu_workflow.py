"""The reference's own integration tests, step for step, against the GPU engine
(/root/reference/tests/test_correct_workflow.py:41-234, tests/test_failing_workflow.py:51-128,
and the CI `cmp report.tsv` lines of .github/workflows/test.yaml:58-77)."""
import tempfile
from pathlib import Path

import pytest

from auriclass_amd.classes import FastaAuriclass, FastqAuriclass
from auriclass_amd.general import check_dependencies, guess_input_type, validate_input_files
from auriclass_amd.main import main
from tests.conftest import GOLDEN, REFDATA

pytestmark = pytest.mark.gpu

COMMON = dict(name="test", output_report_path="tmp_data/test_report.tsv", reference_sketch_path="tests/data/ref_sketch.msh",
              genome_size_range=(40_000, 60_000), kmer_size=27, sketch_size=50_000, minimal_kmer_coverage=3,
              clade_config_path="tests/data/clade_config.csv", non_candida_threshold=0.1, high_dist_threshold=0.003, no_qc=False)


def as_dict(df):
    return {c: {str(i): v for i, v in col.items()} for c, col in df.to_dict().items()}


def test_fastq(refcwd, golden):
    reads = ["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz"]
    validate_input_files(reads)
    check_dependencies()
    s = FastqAuriclass(read_paths=reads, **COMMON)
    with tempfile.TemporaryDirectory() as tmpdir:
        s.query_sketch_path = f"{tmpdir}/tmpfile.msh"
        s.sketch_fastq_query()
        s.run_mash_dist()
        assert as_dict(s.mash_output) == golden["mash_output_to_dict_fastq"]
    s.check_genome_size()
    assert s.estimated_genome_size == 48454.7 and s.qc_genome_size == ""
    s.select_clade()
    assert s.clade == "Lambda phage" and s.minimal_distance == 9.55405e-06
    assert s.closest_sample == "tests/data/NC_001416.1.fasta"
    assert s.check_non_candida() and s.qc_species == ""
    assert s.check_for_outgroup() and s.qc_other_candida == ""
    s.check_high_dist()
    assert s.qc_high_distance == ""
    text = s.get_error_bounds()
    assert text == (GOLDEN / "mash_bounds_k27_p0.99.txt").read_text()
    s.process_error_bounds(text)
    assert s.error_bound == 0.0008979
    s.compare_with_error_bounds()
    assert s.distances == [1.0] and s.samples_within_error_bound == 0 and s.qc_multiple_hits == ""
    s.save_report()


def test_fasta(refcwd, golden):
    s = FastaAuriclass(read_paths=["tests/data/NC_001416.1.fasta.gz"], **COMMON)
    with tempfile.TemporaryDirectory() as tmpdir:
        s.query_sketch_path = f"{tmpdir}/tmpfile.msh"
        s.sketch_fasta_query()
        s.run_mash_dist()
        assert as_dict(s.mash_output) == golden["mash_output_to_dict_fasta"]
    s.parse_genome_size()
    s.check_genome_size()
    assert s.estimated_genome_size == 48502 and s.qc_genome_size == ""
    s.select_clade()
    assert s.clade == "Lambda phage" and s.minimal_distance == 0
    assert s.check_non_candida() and s.check_for_outgroup()
    s.check_high_dist()
    s.process_error_bounds(s.get_error_bounds())
    assert s.error_bound == 0.0008979
    s.compare_with_error_bounds()
    assert s.distances == [1.0] and s.samples_within_error_bound == 0
    s.save_report()


def test_empty_input_files(refcwd):
    s = FastqAuriclass(read_paths=["tests/data/test_empty_1.fq.gz", "tests/data/test_empty_2.fq.gz"], **COMMON)
    check_dependencies()
    with tempfile.TemporaryDirectory() as tmpdir:
        s.query_sketch_path = f"{tmpdir}/tmpfile.msh"
        with pytest.raises(ValueError):
            s.sketch_fastq_query()


def test_non_fastq_or_fasta_and_mixed_inputs(refcwd):
    with pytest.raises(ValueError):
        guess_input_type(["tests/data/ref_sketch.msh"])
    with pytest.raises(ValueError):
        guess_input_type(["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1.fasta.gz"])


def test_cli_reports_are_byte_identical_to_the_reference_reports(refcwd):
    # .github/workflows/test.yaml:58-66
    main(["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz", "-r", "tests/data/ref_sketch.msh",
          "-c", "tests/data/clade_config.csv", "--expected_genome_size", "40000", "60000", "-o", "report.tsv",
          "--log_file_path", "fq.log", "--verbose"])
    assert open("report.tsv", "rb").read() == (REFDATA / "reference_report_fastq.tsv").read_bytes()
    log = open("fq.log").read()
    assert "[mash sketch] Estimated genome size: 48454.7" in log and "[mash dist] mash dist" in log
    # .github/workflows/test.yaml:69-77
    main(["tests/data/NC_001416.1.fasta.gz", "-r", "tests/data/ref_sketch.msh", "-c", "tests/data/clade_config.csv",
          "--expected_genome_size", "40000", "60000", "-o", "report.tsv", "--log_file_path", "fa.log"])
    assert open("report.tsv", "rb").read() == (REFDATA / "reference_report_fasta.tsv").read_bytes()
    # negative invocations must fail (test.yaml:80-106)
    with pytest.raises(FileNotFoundError):
        main(["tests/data/nope.fq.gz", "-r", "tests/data/ref_sketch.msh", "-c", "tests/data/clade_config.csv", "--log_file_path", "x.log"])
    with pytest.raises(ValueError):
        main(["tests/data/ref_sketch.msh", "-r", "tests/data/ref_sketch.msh", "-c", "tests/data/clade_config.csv", "--log_file_path", "x.log"])


def test_mash_named_shim_serves_the_references_subprocess_calls(refcwd):
    """What an unmodified AuriClass checkout does: subprocess.run(["mash", ...], PIPE, PIPE) with
    the shim directory first on PATH (classes.py:92-104, 305-318, 576-596, 696-713; general.py:198-205)."""
    import os
    import subprocess
    from pathlib import Path

    import auriclass_amd

    env = dict(os.environ)
    env["PATH"] = str(Path(auriclass_amd.__file__).parent / "bin") + os.pathsep + env["PATH"]

    def mash(*argv):
        return subprocess.run(["mash", *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)

    assert mash("-h").returncode == 0
    out = mash("sketch", "-r", "-m", "3", "-o", "q.msh", "-k", "27", "-s", "50000",
               "tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz")
    assert "Estimated genome size: 48454.7" in out.stderr.decode()
    rows = mash("dist", "tests/data/ref_sketch.msh", "q.msh").stdout.decode()
    assert rows == ("tests/data/NC_001416.1.fasta\ttests/data/NC_001416.1_1.fq.gz\t9.55405e-06\t0\t48451/48476\n"
                    "tests/data/NC_001604.1.fasta\ttests/data/NC_001416.1_1.fq.gz\t1\t1\t0/50000\n")
    assert mash("bounds", "-k", "27", "-p", "0.99").stdout.decode() == (GOLDEN / "mash_bounds_k27_p0.99.txt").read_text()
    empty = mash("sketch", "-r", "-m", "3", "-o", "e.msh", "-k", "27", "-s", "50000", "tests/data/test_empty_1.fq.gz")
    assert "ERROR: Did not find fasta records in" in empty.stderr.decode() and empty.returncode == 1
    fa = mash("sketch", "-o", "ref2", "-k", "27", "-s", "50000", "tests/data/NC_001416.1.fasta", "tests/data/NC_001604.1.fasta")
    assert fa.returncode == 0 and open("ref2.msh", "rb").read() == (REFDATA / "ref_sketch.msh").read_bytes()


def test_the_binding_stub_printed_in_INTEGRATION_md_works_as_written(refcwd, monkeypatch):
    """INTEGRATION.md shows the ~40-line ctypes module a maintainer would drop next to the reference's
    classes.py; the code block is executed verbatim here and driven with the argv lists of the five
    call sites (classes.py:92-97, 305-312, 576-589, 696-706; general.py:198-205)."""
    import re
    from pathlib import Path

    import auriclass_amd

    root = Path(auriclass_amd.__file__).resolve().parent.parent
    text = (root / "INTEGRATION.md").read_text()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    monkeypatch.setenv("MHX_LIB", str(root / "auriclass_amd" / "lib" / "libmhx.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    run = ns["run"]
    assert run(["mash", "-h"]).stdout == b""
    out = run(["mash", "sketch", "-r", "-m", "3", "-o", "stub_q.msh", "-k", "27", "-s", "50000",
               "tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz"])
    assert b"Estimated genome size: 48454.7" in out.stderr
    rows = run(["mash", "dist", "tests/data/ref_sketch.msh", "stub_q.msh"]).stdout.decode()
    assert rows == ("tests/data/NC_001416.1.fasta\ttests/data/NC_001416.1_1.fq.gz\t9.55405e-06\t0\t48451/48476\n"
                    "tests/data/NC_001604.1.fasta\ttests/data/NC_001416.1_1.fq.gz\t1\t1\t0/50000\n")
    assert run(["mash", "bounds", "-k", "27", "-p", "0.99"]).stdout.decode() == (GOLDEN / "mash_bounds_k27_p0.99.txt").read_text()
    fa = run(["mash", "sketch", "-o", "stub_ref.msh", "-k", "27", "-s", "50000", "tests/data/NC_001416.1.fasta", "tests/data/NC_001604.1.fasta"])
    assert b"Sketching tests/data/NC_001416.1.fasta..." in fa.stderr
    assert open("stub_ref.msh", "rb").read() == (REFDATA / "ref_sketch.msh").read_bytes()
    empty = run(["mash", "sketch", "-r", "-m", "3", "-o", "stub_e.msh", "-k", "27", "-s", "50000", "tests/data/test_empty_1.fq.gz"])
    assert b"ERROR: Did not find fasta records in" in empty.stderr


def test_bench_prints_the_contract_line():
    """`python bench.py` on a small workload: one JSON line with the keys the driver reads, the roofline and CPU-baseline
    objects, and the in-run parity check against the CPU oracle."""
    import json
    import subprocess
    import sys

    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "1", "--reads", "200000",
                          "--cpu-sample-reads", "100000", "--cpu-cores", "2"],
                         capture_output=True, text=True, timeout=600, cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["unit"] == "Gbases/s" and j["higher_is_better"] is True
    assert j["value"] > 1.0 and abs(j["value"] - 200000 * 150 / (j["ms_per_step"] * 1e-3) / 1e9) < 0.01 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 2 and c["value"] > 0
    assert j["parity_on_sample"] is True
