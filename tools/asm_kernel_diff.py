#!/usr/bin/env python3
"""Compare the device assembly of two builds kernel by kernel (`hipcc --cuda-device-only -S` outputs).

    python3 tools/asm_kernel_diff.py before.s after.s

Used when sources are re-organised without the intention to change the product: a kernel whose instruction stream is the same
in both files computes the same bits at the same speed.  Local label numbers (.LBB5_12 -> .LBB*_12), the function ordinal in
.Lfunc_end<N> and comment text are normalised away; everything else (instructions, operands, the .amdhsa_* descriptor of the
kernel) must match.  Prints one line per kernel: SAME / DIFFERENT / only in one file; exit code 1 if any common kernel differs.
"""
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    desc = {}
    cur_desc = None
    for line in open(path, errors="replace"):
        m = re.match(r"\s*\.globl\s+(\S+)", line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            if re.match(r"\.Lfunc_end\d+:", line):
                out[name] = body
                name = None
                continue
            s = line.split(";")[0].rstrip()
            if not s.strip():
                continue
            s = re.sub(r"\.LBB\d+_", ".LBB_", s)
            s = re.sub(r"\.Ltmp\d+", ".Ltmp", s)
            body.append(s)
        else:
            m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
            if m:
                cur_desc = m.group(1)
                desc[cur_desc] = []
            elif cur_desc is not None:
                if ".end_amdhsa_kernel" in line:
                    cur_desc = None
                else:
                    desc[cur_desc].append(line.split(";")[0].strip())
    for k, d in desc.items():
        if k in out:
            out[k] = out[k] + ["<descriptor>"] + d
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    rc = 0
    for k in sorted(set(a) | set(b)):
        if k.startswith("__hip_cuid"):
            continue
        if k not in a or k not in b:
            print(f"{'only before' if k in a else 'only after ':12s} {k}")
        elif a[k] == b[k]:
            print(f"{'SAME':12s} {k}  ({len(a[k])} lines)")
        else:
            n = sum(1 for x, y in zip(a[k], b[k]) if x != y) + abs(len(a[k]) - len(b[k]))
            print(f"{'DIFFERENT':12s} {k}  ({len(a[k])} vs {len(b[k])} lines, {n} differ)")
            rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
