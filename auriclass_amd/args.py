"""Command line of the `auriclass` counterpart: flag for flag the surface of
/root/reference/auriclass/args.py:8-136 (same names, defaults and range checks)."""
from __future__ import annotations

import argparse
from pathlib import Path
from typing import Optional, Sequence

from auriclass_amd.general import check_number_within_range
from auriclass_amd.version import __description__, __package_name__, __version__


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=__description__, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument_group("REQUIRED").add_argument("read_file_paths", nargs="+", help="Paths to read files")

    g = p.add_argument_group("Main arguments")
    g.add_argument("-n", "--name", default="isolate", help="Name of isolate")
    g.add_argument("-o", "--output_report_path", default="report.tsv", type=Path, help="Path to output report")
    g.add_argument("--fastq", action="store_true", help="Input files are fastq files")
    g.add_argument("--fasta", action="store_true", help="Input files are fasta files")
    g.add_argument("--no_qc", action="store_true", dest="no_qc", help="Skip extended QC")
    g.add_argument("--log_file_path", type=Path, help="Path to log file")
    g.add_argument("--verbose", action="store_true", help="Verbose output")
    g.add_argument("--debug", action="store_true", help="Very verbose output")
    g.add_argument("--version", action="version", version=f"{__package_name__} {__version__}")

    q = p.add_argument_group("QC arguments")
    q.add_argument("--expected_genome_size", nargs=2, default=[11_400_000, 14_900_000],
                   type=check_number_within_range(0, 100_000_000),
                   help="Expected genome size range. Defaults 11.4-14.6 Mb are based on 150 NCBI genomes and take "
                        "mash genome size overestimation into account.")
    q.add_argument("--non_candida_threshold", default=0.01, type=check_number_within_range(0, 1),
                   help="If the minimal distance from a reference sample is above this threshold, the sample might "
                        "not be a Candida sp.")
    q.add_argument("--high_dist_threshold", default=0.003, type=check_number_within_range(0, 1),
                   help="If the minimal distance from a reference sample is above this threshold, a warning is "
                        "emitted. See the docs for more info.")

    o = p.add_argument_group("Other arguments\nNOTE: Only change these settings if you are doing something special.\n"
                             "NOTE: This will require rebuilding the reference sketch and recalibration of thresholds!")
    o.add_argument("-r", "--reference_sketch_path", default="", help="Path to reference sketch")
    o.add_argument("-c", "--clade_config_path", default="", help="Path to clade config")
    o.add_argument("-k", "--kmer_size", default=27, type=check_number_within_range(1, 32), help="Kmer size")
    o.add_argument("-s", "--sketch_size", default=50_000, type=check_number_within_range(1000, 1_000_000), help="Sketch size")
    o.add_argument("-m", "--minimal_kmer_coverage", default=3, type=check_number_within_range(1, 100),
                   help="Minimal kmer coverage")
    return p


def auriclass_arg_parser(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    return build_parser().parse_args(argv)
