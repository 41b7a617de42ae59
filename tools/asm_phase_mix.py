#!/usr/bin/env python3
"""Instruction mix between consecutive s_barrier instructions of one kernel in a hipcc -S listing (for the attention kernels:
one entry per phase).  python3 tools/asm_phase_mix.py FILE.s MANGLED_PREFIX [segment_index]"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(n for n, l in enumerate(lines) if re.match(r"^" + re.escape(sys.argv[2]), l))
    end = next(n for n in range(start, len(lines)) if lines[n].startswith(".Lfunc_end"))
    body = [l.strip() for l in lines[start:end]]
    for l in lines[end:end + 60]:
        if re.search(r"\.(num_vgpr|num_agpr|private_seg_size), ", l) or "ScratchSize" in l or "Occupancy" in l:
            print(l.strip())
    bars = [n for n, l in enumerate(body) if l.startswith("s_barrier")]
    segs = list(zip(bars[:-1], bars[1:]))
    for i, (a, b) in enumerate(segs):
        cnt = collections.Counter(l.split()[0] for l in body[a:b] if l and not l.startswith((".", ";")) and not l.endswith(":"))
        print(f"segment {i}: lines {a}..{b}: {sum(cnt.values())} instructions, {sum(v for k, v in cnt.items() if k.startswith('v_mfma'))} MFMAs, "
              f"{sum(v for k, v in cnt.items() if k.startswith('scratch_'))} scratch")
        if len(sys.argv) > 3 and int(sys.argv[3]) == i:
            for k, v in cnt.most_common(40):
                print(f"    {k:34s}{v}")


if __name__ == "__main__":
    main()
