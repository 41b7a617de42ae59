#!/usr/bin/env python3
"""Instruction mix of the largest loops of one kernel in a hipcc -S listing: python3 tools/asm_loop_count.py FILE.s KERNEL_SUBSTRING [n_loops]"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    sub = sys.argv[2]
    start = next(n for n, l in enumerate(lines) if sub in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\w*" + re.escape(sub), l))
    end = next(n for n in range(start, len(lines)) if lines[n].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {m.group(1): n for n, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for n, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            loops.append((n - labels[m.group(1)], labels[m.group(1)], n))
    loops.sort(reverse=True)
    for size, a, b in loops[: int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
        cnt = collections.Counter(l.split()[0] for l in (x.strip() for x in body[a:b]) if l and not l.startswith((".", ";")))
        print(f"loop at lines {a}..{b}: {sum(cnt.values())} instructions")
        for k, v in cnt.most_common(60):
            print(f"  {k:34s}{v}")


if __name__ == "__main__":
    main()
