"""A `mash`-named command for an UNMODIFIED AuriClass checkout: put the directory holding the
`mash` launcher (auriclass_amd/bin) first on PATH and the reference's five subprocess call sites
(/root/reference/auriclass/general.py:198-205 `mash -h`; classes.py:576-596 and 696-706
`mash sketch`; classes.py:92-97 `mash dist`; classes.py:305-312 `mash bounds`) run on the GPU
engine.  Only the argv subsets AuriClass uses are understood; stdout/stderr text and exit
codes follow mash (sketch: exit 1 with 'ERROR: Did not find fasta records in ...')."""
from __future__ import annotations

import sys
from typing import List

from auriclass_amd import engine

USAGE = """
Mash version 2.3 (mhx GPU engine)

Type 'mash --license' for license and copyright information.

Usage:

  mash <command> [options] [arguments ...]

Commands:

  bounds    Print a table of Mash error bounds.

  dist      Estimate the distance of query sequences to references.

  sketch    Create sketches (reduced representations for fast operations).

"""


def _take(args: List[str], flag: str, default=None, cast=str):
    if flag in args:
        i = args.index(flag)
        value = cast(args[i + 1])
        del args[i:i + 2]
        return value
    return default


def main(argv: List[str] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        sys.stdout.write(USAGE)
        return 0
    cmd, args = argv[0], argv[1:]
    try:
        if cmd == "sketch":
            reads = "-r" in args
            if reads:
                args.remove("-r")
            m = _take(args, "-m", 1, int)
            out = _take(args, "-o", None)
            k = _take(args, "-k", 21, int)
            s = _take(args, "-s", 1000, int)
            if out is None or not args:
                sys.stderr.write("ERROR: mash sketch needs -o <out> and at least one input\n")
                return 1
            if not out.endswith(".msh"):
                out += ".msh"
            try:
                text, _ = engine.sketch_files(args, k, s, out, reads=reads, min_mult=m if reads else 1)
            except engine.NoRecordsError as exc:
                sys.stderr.write("\n" + exc.message + "\n")
                return 1
            sys.stderr.write(text)
            return 0
        if cmd == "dist":
            if len(args) != 2:
                sys.stderr.write("ERROR: mash dist <reference> <query>\n")
                return 1
            sys.stdout.write(engine.dist_files(args[0], args[1]))
            return 0
        if cmd == "bounds":
            k = _take(args, "-k", 21, int)
            p = _take(args, "-p", 0.99, float)
            sys.stdout.write(engine.bounds(k, p))
            return 0
    except engine.EngineError as exc:
        sys.stderr.write(exc.message + "\n")
        return 1
    sys.stderr.write(f"ERROR: unsupported mash command for the mhx shim: {cmd}\n")
    return 1


if __name__ == "__main__":
    sys.exit(main())
