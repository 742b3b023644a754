"""Extract the golden CONSTANTS (data, not code) that the reference's own tests pin
for the sketch/dist/bounds path, and store them as JSON/text fixtures.

Run in the build container only (needs /root/reference):
    python tests/golden/make_reference_constants.py
Sources: /root/reference/tests/test_correct_workflow.py:18-38 (dist tables, bounds text),
:99/:105/:129/:197 (scalars); tests/data/* copied verbatim into tests/golden/refdata/.
"""
import ast
import json
from pathlib import Path

REF = Path("/root/reference/tests/test_correct_workflow.py")
OUT = Path(__file__).resolve().parent

tree = ast.parse(REF.read_text())
consts = {}
for node in tree.body:
    if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
        name = node.targets[0].id
        if name in ("mash_output_to_dict_fastq", "mash_output_to_dict_fasta", "error_bounds_text_ref"):
            consts[name] = ast.literal_eval(node.value)

(OUT / "mash_bounds_k27_p0.99.txt").write_text(consts.pop("error_bounds_text_ref"))
# JSON keys must be str; keep row index as str
for k, v in consts.items():
    consts[k] = {col: {str(i): val for i, val in rows.items()} for col, rows in v.items()}
consts["scalars"] = {
    "fastq_estimated_genome_size": 48454.7,   # test_correct_workflow.py:99
    "fastq_minimal_distance": 9.55405e-06,    # :105
    "error_bound_s50000": 0.0008979,          # :129
    "fasta_estimated_genome_size": 48502,     # :197
    "fasta_minimal_distance": 0,              # :203
}
(OUT / "reference_constants.json").write_text(json.dumps(consts, indent=1) + "\n")
print("wrote", sorted(p.name for p in OUT.iterdir()))
