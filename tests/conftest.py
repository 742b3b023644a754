import json
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"
REFDATA = GOLDEN / "refdata"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return json.loads((GOLDEN / "reference_constants.json").read_text())


@pytest.fixture()
def refcwd(tmp_path, monkeypatch):
    """A scratch cwd in which `tests/data/...` resolves to the reference's fixture files,
    so that path strings (which end up inside .msh files and dist rows) are the very
    strings the reference's tests use (tests/test_correct_workflow.py:49-57)."""
    import gzip

    d = tmp_path / "tests" / "data"
    d.mkdir(parents=True)
    for f in REFDATA.iterdir():
        os.symlink(f, d / f.name)
    # the reference sketch was made from the uncompressed FASTA files (names end in .fasta)
    for stem in ("NC_001416.1.fasta", "NC_001604.1.fasta"):
        (d / stem).write_bytes(gzip.decompress((REFDATA / (stem + ".gz")).read_bytes()))
    (tmp_path / "tmp_data").mkdir()
    monkeypatch.chdir(tmp_path)
    return tmp_path
