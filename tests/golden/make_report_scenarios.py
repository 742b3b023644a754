"""Generate golden QC/report scenarios from the REFERENCE's own Python layer (data, not code).

Run in the build container only (needs /root/reference; pyfastx is replaced by a stub that only
reports a FASTA size, and subprocess.run by a function that returns scripted mash texts):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_report_scenarios.py
Each scenario = the mash-shaped texts fed in (dist rows for 24 references, the k=27 bounds table,
an estimated genome size) plus the settings; the stored expectation is the reference's one-row
report (or the exception type it raised).  tests/test_report_plumbing.py replays them through
auriclass_amd.classes.  Anchors: /root/reference/auriclass/classes.py:121-538, 618-660, 753-795.
"""
import json
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
stub = types.ModuleType("pyfastx")


class _Fasta:
    size = 0

    def __init__(self, *a, **k):
        pass


stub.Fasta = _Fasta
stub.Fastq = _Fasta
sys.modules["pyfastx"] = stub
sys.path.insert(0, "/root/reference")
import auriclass.classes as rc  # noqa: E402

BOUNDS = (OUT / "mash_bounds_k27_p0.99.txt").read_text()
CLADES = ["Clade I", "Clade II", "Clade III", "Clade IV", "Clade V", "outgroup"]


class _Done:
    def __init__(self, out=b"", err=b""):
        self.stdout, self.stderr, self.returncode = out, err, 0


def scenario(rng, i):
    mode = "fastq" if rng.random() < 0.5 else "fasta"
    refs = ["refs/ref_%02d.fasta" % j for j in range(24)]
    clade_of = {r: CLADES[min(j // 4, 5)] for j, r in enumerate(refs)}
    kind = int(rng.integers(0, 6))
    d = rng.uniform(0.02, 0.2, 24)                          # far from everything
    best = int(rng.integers(0, 24))
    if kind <= 2:                                            # a clear closest reference
        d[best] = float(rng.choice([0.0, 1e-5, 4e-4, 0.002, 0.0031, 0.009, 0.0101]))
    if kind == 1:                                            # a second reference within the error bound
        other = int(rng.integers(0, 24))
        d[other] = d[best] + float(rng.choice([0.0, 1e-4, 5e-4, 0.002]))
    if kind == 3:                                            # closest is an outgroup member
        best = int(rng.integers(20, 24))
        d[best] = float(rng.choice([0.0, 0.002, 0.02]))
    if kind == 4:                                            # ties
        d[:] = 0.05
        d[int(rng.integers(0, 24))] = 0.001
        d[int(rng.integers(0, 24))] = 0.001
    if kind == 5:                                            # everything far: not Candida
        d = rng.uniform(0.011, 1.0, 24)
        d[int(rng.integers(0, 24))] = 1.0
    query = "reads_1.fq.gz" if mode == "fastq" else "asm.fasta"
    rows = []
    for r, dist in zip(refs, d):
        shared = int(round((1 - min(dist, 1.0) * 10) * 50000)) if dist < 0.1 else 0
        rows.append("%s\t%s\t%g\t%g\t%d/50000" % (r, query, dist, 0 if dist < 0.05 else 1, max(shared, 0)))
    est = float(rng.choice([9.0e6, 11.4e6, 12.3e6, 14.9e6, 15.0e6, 20e6]))
    size_range = [int(x) for x in rng.choice([[11_400_000, 14_900_000], [12_000_000, 13_000_000], [1, 100_000_000]])]
    return {
        "id": i, "mode": mode, "dist_text": "\n".join(rows) + "\n", "clades": clade_of,
        "estimated_genome_size": est, "genome_size_range": size_range,
        "non_candida_threshold": float(rng.choice([0.01, 0.005, 0.05])),
        "high_dist_threshold": float(rng.choice([0.003, 0.001, 0.01])),
        "no_qc": bool(rng.random() < 0.15), "sketch_size": int(rng.choice([50000, 50000, 10000, 1000])),
    }


def run_reference(sc, tmp):
    clade_csv = tmp / "clades.csv"
    clade_csv.write_text("filename,clade\n" + "".join("%s,%s\n" % kv for kv in sc["clades"].items()))

    def fake_run(argv, stdout=None, stderr=None, **kw):
        cmd = argv[1]
        if cmd == "sketch":
            return _Done(err=("Estimated genome size: %g\nEstimated coverage:    40\n" % sc["estimated_genome_size"]).encode())
        if cmd == "dist":
            return _Done(out=sc["dist_text"].encode())
        if cmd == "bounds":
            return _Done(out=BOUNDS.encode())
        return _Done()

    rc.subprocess.run = fake_run
    _Fasta.size = int(sc["estimated_genome_size"])
    cls = rc.FastqAuriclass if sc["mode"] == "fastq" else rc.FastaAuriclass
    report = tmp / "report.tsv"
    obj = cls(name="isolate", output_report_path=report,
              read_paths=[Path("reads_1.fq.gz"), Path("reads_2.fq.gz")] if sc["mode"] == "fastq" else [Path("asm.fasta")],
              reference_sketch_path=Path("refs.msh"), kmer_size=27, sketch_size=sc["sketch_size"], minimal_kmer_coverage=3,
              clade_config_path=clade_csv, genome_size_range=sc["genome_size_range"],
              non_candida_threshold=sc["non_candida_threshold"], high_dist_threshold=sc["high_dist_threshold"], no_qc=sc["no_qc"])
    try:
        obj.run()
    except Exception as e:      # the reference's quirks are part of the contract
        return {"raises": type(e).__name__}
    return {"report": report.read_text(), "clade": obj.clade, "minimal_distance": float(obj.minimal_distance),
            "samples_within_error_bound": int(obj.samples_within_error_bound), "error_bound": float(obj.error_bound)}


def main():
    rng = np.random.default_rng(2026)
    out = []
    with tempfile.TemporaryDirectory() as t:
        for i in range(80):
            sc = scenario(rng, i)
            sc["expect"] = run_reference(sc, Path(t))
            out.append(sc)
        # the quirks of SURVEY.md appendix A: a sketch size without a row in the bounds table, and a closest
        # distance beyond the last bounds column (reachable only with a huge non-Candida threshold)
        for j, (ss, thr, dmin) in enumerate([(2000, 0.01, 0.001), (50000, 0.5, 0.45), (100, 0.01, 0.0), (500000, 0.01, 0.002)]):
            sc = scenario(rng, 80 + j)
            sc["sketch_size"], sc["non_candida_threshold"], sc["no_qc"] = ss, thr, False
            rows = sc["dist_text"].splitlines()
            f = rows[0].split("\t")
            f[2] = "%g" % dmin
            rows = ["\t".join(f)] + ["\t".join(r.split("\t")[:2] + ["0.6", "1", "0/50000"]) for r in rows[1:]]
            sc["dist_text"] = "\n".join(rows) + "\n"
            sc["expect"] = run_reference(sc, Path(t))
            out.append(sc)
    (OUT / "report_scenarios.json").write_text(json.dumps(out, indent=0) + "\n")
    kinds = {}
    for sc in out:
        key = sc["expect"].get("raises") or sc["expect"]["report"].splitlines()[1].split("\t")[3]
        kinds[key] = kinds.get(key, 0) + 1
    print(len(out), "scenarios:", kinds)


if __name__ == "__main__":
    main()
