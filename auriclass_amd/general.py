"""Input validation and format sniffing: the counterpart of /root/reference/auriclass/general.py
with the two external dependencies replaced -- `pyfastx` probes by the engine's sniffers
(general.py:68-115) and the `mash -h` probe by loading the GPU engine (general.py:181-205)."""
from __future__ import annotations

import argparse
import logging
from pathlib import Path
from typing import Callable, List

from auriclass_amd import engine


def add_tag(tag: str, lines: str) -> str:
    """Prefix every non-empty line of `lines` with "[tag]" (a bare "[tag]" for empty input)."""
    prefix = f"[{tag}]"
    if not lines:
        return prefix
    return "\n".join(f"{prefix} {ln}" for ln in lines.split("\n") if ln != "")


def check_number_within_range(minimum: float = 0, maximum: float = 1) -> Callable[[str], str]:
    """argparse `type=` factory: accepts a number inside [minimum, maximum] and hands the
    ORIGINAL STRING back (the reference does, general.py:57-63; main() casts later)."""

    def _validator(value: str) -> str:
        number = float(value)
        if number < minimum or number > maximum:
            raise argparse.ArgumentTypeError(
                f"Supplied value {value} is not within expected range {minimum} to {maximum}."
            )
        return str(value)

    return _validator


def is_fastq(file: str) -> bool:
    return engine.sniff_fastq(file)


def is_fasta(file: str) -> bool:
    return engine.sniff_fasta(file)


def validate_input_files(list_of_files: List[str]) -> None:
    for candidate in list_of_files:
        if not Path(candidate).exists():
            raise FileNotFoundError(f"Required input file {candidate} does not exist")


def validate_argument_logic(args: argparse.Namespace) -> argparse.Namespace:
    low, high = (float(v) for v in args.expected_genome_size[:2])
    args.expected_genome_size = [low, high]
    if low > high:
        raise ValueError("Expected genome size range is invalid: lower bound is higher than upper bound")
    if low < 100 and high < 100:
        logging.warning(
            f"Expected genome size range boundaries {args.expected_genome_size} are below 100: treating these as Mbp instead of bp"
        )
        args.expected_genome_size = [low * 1_000_000, high * 1_000_000]
    return args


def check_dependencies() -> None:
    """The reference probes `mash -h` and raises FileNotFoundError when the binary is absent.
    Here the dependency is the GPU engine: same exception type when it cannot be used."""
    try:
        engine.load()
        engine.init()
    except engine.EngineError as exc:
        raise FileNotFoundError(f"The mhx GPU engine is not available: {exc.message}") from exc


def guess_input_type(list_of_file_paths: List[str]) -> str:
    n_fastq = n_fasta = 0
    for path in list_of_file_paths:
        fq, fa = is_fastq(path), is_fasta(path)
        if fq and fa:
            raise ValueError(
                f"Input file {path} can be parsed as both fastq and fasta. Please specify --fastq or --fasta"
            )
        if fq:
            n_fastq += 1
        elif fa:
            n_fasta += 1
        else:
            raise ValueError(f"Input file {path} is not a fastq or fasta file")
    if n_fastq and n_fasta:
        raise ValueError("Input files are a mix of fastq and fasta files")
    if n_fastq:
        return "fastq"
    if n_fasta:
        return "fasta"
    raise ValueError("No input files were found")


def confirm_input_type(list_of_file_paths: List[str], input_type: str) -> None:
    """Only warns (the reference does not raise here, general.py:282-292)."""
    probe = {"fastq": is_fastq, "fasta": is_fasta}.get(input_type)
    if probe is None:
        return
    for path in list_of_file_paths:
        if not probe(path):
            logging.warning(
                f"Input file {path} cannot be parsed as a {input_type} file, please check if --{input_type} is appropriate"
            )
