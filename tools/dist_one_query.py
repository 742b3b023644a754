import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from auriclass_amd import engine
engine.init(0)
rng = np.random.default_rng(1)
s = 50000
top = 1 << 54
base = np.unique(rng.integers(0, top, size=s, dtype=np.uint64))
R = np.zeros((24, s), np.uint64); rl = np.zeros(24, np.uint32)
for j in range(24):
    v = np.unique(np.where(rng.random(len(base)) < 0.02 * (j + 1), rng.integers(0, top, size=len(base), dtype=np.uint64), base))
    R[j, :len(v)] = v; rl[j] = len(v)
Q = R[3:4].copy(); ql = rl[3:4].copy()
for it in range(4):
    c, d, x = engine.dist_batch(Q, ql, R, rl, 27, s)
    print(os.environ.get("MHX_DIST_GENERIC", "default"), "kernel ms", round(engine.load().mhx_last_dist_kernel_ms(), 4), c[0, :4], d[0, :4])
