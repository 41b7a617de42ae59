#!/usr/bin/env python3
"""Per-instruction energy of the attention loop's ingredients (GPU box; build/attn_energy is cross-compiled in the CPU container:
hipcc --offload-arch=gfx950 -O3 -o build/attn_energy tools/probes/attn_energy.hip).

Every variant of tools/probes/attn_energy.hip runs for a few seconds on random operands while package power and shader clock
are read from rocm-smi once a second.  With the package at its power limit,
    energy per MFMA slot (one wave-level v_mfma_f32_16x16x32_bf16 + its share of the other instructions) = power x time / slots,
and the difference between two variants divided by the number of added instructions is the energy of ONE such wave instruction.
Printed: a markdown table per variant, the per-instruction energies, and the bound they give for kernel 3's mix (and for the
ORMAX / 64-rows-per-wave mixes):  TFLOP/s <= power x 16384 FLOP / energy per slot."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "attn_energy")


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    except Exception:      # noqa: BLE001
        return None, None
    p = re.search(r"Package Power \(W\): ([\d.]+)", out)
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None), (float(c.group(1)) if c else None)


def run_variant(v, seconds, data="random", one_wave=False):
    proc = subprocess.Popen([EXE, str(v), str(seconds), data] + (["1"] if one_wave else []), stdout=subprocess.PIPE, text=True)
    samples = []
    t0 = time.time()
    while proc.poll() is None:
        time.sleep(1.0)
        if 1.5 < time.time() - t0 < seconds - 0.3:          # skip the ramp at both ends
            samples.append(smi())
    rec = json.loads(proc.stdout.read().strip().splitlines()[-1])
    pw = [p for p, _ in samples if p]
    ck = [c for _, c in samples if c]
    rec["power_w"] = sum(pw) / len(pw) if pw else None
    rec["sclk_mhz"] = sum(ck) / len(ck) if ck else None
    return rec


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    names = subprocess.run([EXE], capture_output=True, text=True).stdout.strip().splitlines()
    cap = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout
    m = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", cap)
    cap_w = float(m.group(1)) if m else None
    print(f"package power limit: {cap_w} W; {seconds:.0f} s per variant, random operands, 2 waves per SIMD on every CU\n")
    recs = [run_variant(i, seconds) for i in range(len(names))]
    zero = run_variant(0, seconds, "zeros")
    extra = [zero, run_variant(7, seconds, "zeros"), run_variant(0, seconds, one_wave=True), run_variant(7, seconds, one_wave=True),
             run_variant(9, seconds, one_wave=True), run_variant(9, seconds, "zeros", one_wave=True)]
    print("| variant (per 32 MFMAs and wave) | TFLOP/s | power W | sclk MHz | nJ per MFMA slot |\n|---|---|---|---|---|")
    for r in recs + extra:
        p = r["power_w"] or cap_w
        r["nj_slot"] = p * r["seconds"] / r["mfma_slots"] * 1e9
        print(f"| {r['name']}{' [ZERO operands]' if r['data'] == 'zeros' else ''}{' [ONE wave per SIMD]' if r.get('waves_per_simd') == 1 else ''} | {r['tflops']:.0f} | {r['power_w'] and round(r['power_w'])} | "
              f"{r['sclk_mhz'] and round(r['sclk_mhz'])} | {r['nj_slot']:.2f} |")
    e = {r["variant"]: r["nj_slot"] for r in recs}
    per = {"v_mfma_f32_16x16x32_bf16": e[0], "ds_read_b128 (at 1 per 2 MFMAs)": (e[1] - e[0]) * 2, "ds_read_b128 (at 1 per 4 MFMAs)": (e[2] - e[0]) * 4,
           "v_exp_f32": (e[3] - e[0]) * 2, "v_cvt_pk_bf16_f32": (e[4] - e[3]) * 4, "v_max3_f32": (e[5] - e[4]) * 4, "v_or3_b32": (e[6] - e[4]) * 8}
    print("\n| instruction (one wave-level issue, 64 lanes) | energy nJ |\n|---|---|")
    for k, v in per.items():
        print(f"| {k} | {v:.2f} |")
    p = cap_w or 1400.0
    print("\n| mix | sum of parts nJ per slot | measured nJ per slot | bound at the power limit, TFLOP/s (algorithmic: x 32/34 for the row-sum MFMAs) |\n|---|---|---|---|")
    parts3 = e[0] + 0.5 * per["ds_read_b128 (at 1 per 2 MFMAs)"] + 0.5 * per["v_exp_f32"] + 0.25 * per["v_cvt_pk_bf16_f32"] + 0.25 * per["v_max3_f32"]
    partso = parts3 - 0.25 * per["v_max3_f32"] + 0.125 * per["v_or3_b32"]
    parts4 = parts3 - 0.5 * per["ds_read_b128 (at 1 per 2 MFMAs)"] + 0.25 * per["ds_read_b128 (at 1 per 4 MFMAs)"]
    for name, parts, meas in (("kernel 3 (max3, 1 read per 2)", parts3, e[7]), ("ORMAX (or3, 1 read per 2)", partso, e[8]),
                              ("64 query rows per wave (max3, 1 read per 4)", parts4, e[9])):
        print(f"| {name} | {parts:.2f} | {meas:.2f} | {p * 16384 / meas / 1e3:.0f} ({p * 16384 / meas / 1e3 * 32 / 34:.0f}) |")
    print("\n" + json.dumps({"cap_w": cap_w, "variants": recs, "extra": extra, "per_instruction_nj": per}))


if __name__ == "__main__":
    main()
