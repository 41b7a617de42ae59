#!/usr/bin/env python3
"""GPU idle time between kernels inside the denoising loop, from a rocprofv3 rocpd database of bench.py
(`rocprofv3 --kernel-trace -d DIR -o NAME -- python3 bench.py ...` writes DIR/NAME_results.db):

    python3 tools/rocpd_gaps.py DB [first_attn_launch [n_attn_launches]]

The window runs from the start of self-attention launch `first` to the start of launch `first + n` (defaults 120 / 360: past the
warm-up step, four-odd steps long).  Prints window length, union of the kernel intervals, idle share, and the idle time grouped by
the (kernel before the gap -> kernel after it) pair."""
import collections
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 360
    attn = [r[0] for r in db.execute("select start from kernels where name like '%flash_attn_fwd_kernel3%' order by start")]
    if len(attn) <= first + n:
        sys.exit(f"only {len(attn)} self-attention launches in the trace")
    t0, t1 = attn[first], attn[first + n]
    rows = list(db.execute("select name, start, end from kernels where start >= ? and start < ? order by start", (t0, t1)))
    busy, cur_end, gaps, prev = 0, t0, collections.Counter(), None
    counts = collections.Counter()
    for name, s, e in rows:
        if s > cur_end:
            if prev is not None:
                gaps[(short(prev), short(name))] += s - cur_end
                counts[(short(prev), short(name))] += 1
            busy += e - s
            cur_end, prev = e, name
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end, prev = e, name
    win = t1 - t0
    idle = sum(gaps.values())
    print(f"window {win / 1e6:.1f} ms, {len(rows)} launches, {n} self-attention launches")
    print(f"kernels busy {busy / 1e6:.1f} ms = {100 * busy / win:.2f} %; idle between kernels {idle / 1e6:.1f} ms = {100 * idle / win:.2f} %\n")
    print("| before the gap | after the gap | gaps | idle ms | mean us | % of window |\n|---|---|---|---|---|---|")
    for k, v in gaps.most_common(25):
        print(f"| {k[0]} | {k[1]} | {counts[k]} | {v / 1e6:.2f} | {v / 1e3 / counts[k]:.1f} | {100 * v / win:.3f} |")


if __name__ == "__main__":
    main()
