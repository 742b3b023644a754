"""CPU-baseline worker (test/bench infrastructure): sketch one record shard of a FASTQ file with the
C oracle and save the partial sketch.   python -m oracle.shard_worker FILE LO HI K S OUT.npy"""
import sys

import numpy as np

from oracle import mash_oracle as mo


def main() -> None:
    path, lo, hi, k, s, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    data = np.memmap(path, dtype=np.uint8, mode="r")[lo:hi].tobytes()
    ref = mo.Sketcher(k, s, 1)
    ref.add_fastx(data)
    h, _ = ref.finish()
    np.save(out, h)


if __name__ == "__main__":
    main()
