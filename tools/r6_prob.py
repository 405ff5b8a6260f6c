import numpy as np
def problem(n=20000, deg=15, seed=0):
    rng = np.random.default_rng(seed)
    src = np.repeat(np.arange(n), deg)
    dst = (src + rng.integers(1, 200, n * deg)) % n
    neg = rng.integers(0, n, n * deg)
    e = np.concatenate([np.stack([src, dst], 1), np.stack([src, neg], 1)])
    e = e[e[:, 0] != e[:, 1]]
    e = np.unique(np.stack([e.min(1), e.max(1)], 1), axis=0)
    w = np.ones(e.shape[0], dtype=np.float32)
    w[rng.permutation(e.shape[0])[: e.shape[0] // 2]] = -1.0
    return e, w
