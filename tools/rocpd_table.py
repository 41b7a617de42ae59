#!/usr/bin/env python3
"""Per-(kernel, grid) table from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes
DIR/NAME_results.db): python3 tools/rocpd_table.py DB [top] -> markdown on stdout."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = list(db.execute("select name, grid_x, count(*), sum(end-start)/1e6, avg(end-start)/1e6 from kernels "
                           "group by name, grid_x order by 4 desc"))
    tot = sum(r[3] for r in rows)
    print(f"total kernel time {tot:.1f} ms over {sum(r[2] for r in rows)} launches\n")
    print("| kernel | grid (threads) | calls | total ms | avg ms | % |\n|---|---|---|---|---|---|")
    for r in rows[:top]:
        print(f"| {r[0][:90]} | {r[1]} | {r[2]} | {r[3]:.2f} | {r[4]:.4f} | {100 * r[3] / tot:.2f} |")


if __name__ == "__main__":
    main()
