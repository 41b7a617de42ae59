#!/usr/bin/env python3
"""Per-kernel counter table from a rocprofv3 --pmc rocpd database: python3 tools/rocpd_pmc.py DB [name1|name2...] -> markdown.
A dispatch has one row per counter instance (XCD / shader engine): values are SUMMED per dispatch, then averaged over the
dispatches of a kernel.  GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 = cycles of the dispatch."""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    subs = sys.argv[2].split("|") if len(sys.argv) > 2 else [""]
    per = collections.defaultdict(lambda: [0.0, 0.0, 0])      # (kernel, counter, dispatch) -> [sum, duration, instances]
    for name, cn, disp, val, dur in db.execute("select name, counter_name, dispatch_id, counter_value, duration from pmc_events"):
        e = per[(name, cn, disp)]
        e[0] += val
        e[1] = dur
        e[2] += 1
    agg = collections.defaultdict(list)
    for (name, cn, disp), (v, dur, inst) in per.items():
        agg[(name, cn)].append((v, dur / 1e6, inst))
    print("| kernel | counter | dispatches | instances per dispatch | mean sum per dispatch | mean duration ms |\n|---|---|---|---|---|---|")
    for (name, cn), rows in sorted(agg.items()):
        if any(s in name for s in subs):
            n = len(rows)
            print(f"| {name[:70]} | {cn} | {n} | {rows[0][2]} | {sum(r[0] for r in rows) / n:.6g} | {sum(r[1] for r in rows) / n:.4f} |")


if __name__ == "__main__":
    main()
