"""CPU simulation (numpy) of a bank-class-aware iteration scheduler for the LDS-ring kernel: how much
padding would it cost to keep every 64-entry iteration at <= 2 entries per LDS bank class on the row
side AND on the column side (then a 2-colouring of the entries gives two conflict-free 32-lane
passes for every access: 2.6 instead of ~6 LDS clocks, tools/bankprobe)?  Config-4 statistics of one
consumer wave: 326 rows, ~32 half-edges per 1024-column chunk, 977 chunks, window of SPAN chunks.
Not part of the product; an estimate for the next round."""
import sys
import numpy as np

rng = np.random.default_rng(0)
ROWS, NCHUNK, PER_CHUNK, CLASSES = 326, 977, 32.0, 32


def cost(classes):
    """LDS clocks of one access under the measured rule (two 32-lane passes, deepest class each) when
    the entries are dealt to the two passes class by class (what lane placement can do)."""
    cnt = np.bincount(classes, minlength=CLASSES)
    deep = cnt.max() if len(classes) else 0
    return int(np.ceil(deep / 2)) + int(np.floor(deep / 2)) if deep else 0


def run(span, cap, defer_dups=True):
    # stream of (chunk, row, col_class)
    n = rng.poisson(PER_CHUNK, NCHUNK)
    chunk = np.repeat(np.arange(NCHUNK), n)
    row = rng.integers(0, ROWS, len(chunk))
    colc = rng.integers(0, CLASSES, len(chunk))
    rowc = row % CLASSES
    pending = list(range(len(chunk)))  # indices in stream order
    iters = pad = 0
    rcost = ccost = 0
    pos = 0
    queue = []  # carried entries (stream order)
    while pos < len(chunk) or queue:
        # window: oldest needed chunk m .. m + span - 1
        m = chunk[queue[0]] if queue else chunk[pos]
        while pos < len(chunk) and chunk[pos] < m + span:
            queue.append(pos)
            pos += 1
        take, keep = [], []
        rc = np.zeros(CLASSES, int)
        cc = np.zeros(CLASSES, int)
        rows = set()
        for e in queue:
            ok = len(take) < 64 and rc[rowc[e]] < cap and cc[colc[e]] < cap and (row[e] not in rows)
            if ok:
                take.append(e)
                rc[rowc[e]] += 1
                cc[colc[e]] += 1
                rows.add(row[e])
            else:
                keep.append(e)
        queue = keep
        iters += 1
        pad += 64 - len(take)
        t = np.array(take, int)
        rcost += cost(rowc[t])
        ccost += cost(colc[t])
    total = len(chunk)
    return iters, 100.0 * pad / (64 * iters), rcost / iters, ccost / iters, total


if __name__ == "__main__":
    print("cap = max entries per bank class and side; cost = LDS passes per access (x ~1.3 clocks)")
    for span in (6, 8):
        for cap in (64, 4, 3, 2):
            it, padp, rcst, ccst, total = run(span, cap)
            print("window %d chunks, cap %2d: %5d iterations for %d entries, %5.1f %% padding, row-side %.2f passes, "
                  "column-side %.2f passes" % (span, cap, it, total, padp, rcst, ccst))
