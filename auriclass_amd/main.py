#!/usr/bin/env python3
"""`auriclass` entry point on the GPU engine: same flow as /root/reference/auriclass/main.py:19-118
(logging setup, default data discovery, validation, type guess, one FastqAuriclass or
FastaAuriclass, run())."""
from __future__ import annotations

import logging
from datetime import datetime
from pathlib import Path
from typing import Optional, Sequence

from auriclass_amd.args import auriclass_arg_parser
from auriclass_amd.classes import FastaAuriclass, FastqAuriclass
from auriclass_amd.general import (
    check_dependencies,
    confirm_input_type,
    guess_input_type,
    validate_argument_logic,
    validate_input_files,
)


def _default_data_file(filename: str):
    """Installed layout: <prefix>/lib/.../auriclass_amd -> <prefix>/data/<filename> (main.py:38-52)."""
    for parent in Path(__file__).parents:
        if parent.stem == "lib":
            return parent.parent.joinpath("data", filename)
    return ""


def main(argv: Optional[Sequence[str]] = None) -> None:
    args = auriclass_arg_parser(argv)

    log_path = args.log_file_path or args.output_report_path.with_suffix(
        f".{datetime.now().strftime('%Y-%m-%d_%H-%M-%S')}.log"
    )
    logging.basicConfig(filename=log_path, filemode="w", format="%(asctime)s %(levelname)s %(message)s",
                        datefmt="%H:%M:%S", force=True)
    logging.getLogger().addHandler(logging.StreamHandler())

    if args.reference_sketch_path == "":
        args.reference_sketch_path = _default_data_file("Candida_auris_clade_references.msh")
    if args.clade_config_path == "":
        args.clade_config_path = _default_data_file("clade_config.csv")

    if args.verbose:
        logging.getLogger().setLevel(logging.INFO)
    if args.debug:
        logging.getLogger().setLevel(logging.DEBUG)

    validate_input_files(args.read_file_paths)
    validate_input_files([args.reference_sketch_path])
    validate_input_files([args.clade_config_path])
    args = validate_argument_logic(args)
    check_dependencies()

    if args.fastq or args.fasta:
        input_type = "fastq" if args.fastq else "fasta"
        confirm_input_type(args.read_file_paths, input_type)
    else:
        input_type = guess_input_type(args.read_file_paths)

    sample_class = FastqAuriclass if input_type == "fastq" else FastaAuriclass
    sample = sample_class(
        name=args.name,
        output_report_path=args.output_report_path,
        read_paths=args.read_file_paths,
        reference_sketch_path=args.reference_sketch_path,
        kmer_size=int(args.kmer_size),
        sketch_size=int(args.sketch_size),
        minimal_kmer_coverage=int(args.minimal_kmer_coverage),
        clade_config_path=args.clade_config_path,
        genome_size_range=[int(size) for size in args.expected_genome_size],
        non_candida_threshold=float(args.non_candida_threshold),
        high_dist_threshold=float(args.high_dist_threshold),
        no_qc=args.no_qc,
    )
    sample.run()


if __name__ == "__main__":
    main()
