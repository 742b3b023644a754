"""Wall time of ONE AuriClass process per sample (the reference's unit of work, /root/reference/auriclass/main.py): the CLI
mirror started as a child process on the reference's own FASTA fixture.  ~0.6 s, of which the interpreter and its imports
(pandas) take ~0.35 s and the HIP runtime 0.2-0.3 s to come up (tools/startup_time.py); sketch + dist + report ~0.03 s.
(Tried: bringing the engine up on a helper thread behind the imports -- 602 -> 575 ms only: the runtime's own library
loading and the imports' serialise on the dynamic loader; not kept.)  Run from the repo root."""
import os, statistics, subprocess, sys, tempfile, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = os.path.join(root, "tests", "golden", "refdata")
fa = os.path.join(data, "NC_001416.1.fasta.gz")
ref = os.path.join(data, "ref_sketch.msh")
cfg = os.path.join(data, "clade_config.csv")
with tempfile.TemporaryDirectory() as d:
    for label, env in (("one process per sample", {}),):
        ts = []
        for i in range(7):
            out = os.path.join(d, f"r{i}.tsv")
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "auriclass_amd.main", fa, "-r", ref, "-c", cfg, "-o", out, "--log_file_path", os.path.join(d, "log"),
                                "--expected_genome_size", "40000", "60000"], cwd=root, env=dict(os.environ, **env), capture_output=True)
            ts.append(time.perf_counter() - t0)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
        print(f"{label:24s}: process wall {statistics.median(ts[2:]) * 1e3:.0f} ms median, {min(ts) * 1e3:.0f} min", flush=True)
    print(open(os.path.join(d, "r0.tsv")).read())
