import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from auriclass_amd import engine
engine.init(0)
rng = np.random.default_rng(1)
s = int(os.environ.get("S", "50000"))
top = 1 << 54
base = np.unique(rng.integers(0, top, size=s, dtype=np.uint64))
R = np.zeros((24, s), np.uint64); rl = np.zeros(24, np.uint32)
for j in range(24):
    v = np.unique(np.where(rng.random(len(base)) < 0.02 * (j + 1), rng.integers(0, top, size=len(base), dtype=np.uint64), base))
    R[j, :len(v)] = v; rl[j] = len(v)
Q = R[3:4].copy(); ql = rl[3:4].copy()
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")))
from oracle import mash_oracle as mo
for it in range(4):
    c, d, x = engine.dist_batch(Q, ql, R, rl, 27, s)
    ok = all(mo.compare(R[j, :rl[j]], Q[0, :ql[0]], s, 27)[:2] == (int(c[0, j]), int(d[0, j])) for j in (0, 3, 11, 23))
    print(os.environ.get("MHX_DIST_GENERIC", "default"), "s", s, "oracle ok", ok, "fallback blocks", engine.load().mhx_last_dist_fallback_blocks(), "kernel ms", round(engine.load().mhx_last_dist_kernel_ms(), 4), c[0, :4], d[0, :4])
