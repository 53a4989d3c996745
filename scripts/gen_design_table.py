#!/usr/bin/env python3
"""Regenerate the current-state table of DESIGN.md section 3.4 from the committed measurements -- and from nothing else:

    profiles/r<NN>_bench_default_n1.json           the default `python bench.py` line (headline + `secondary` + `secondary_vectors` + cpu_baseline)
    profiles/r<NN>_other_kernels.jsonl             scripts/bench_other_kernels.py (residual / energy / source / SpMV / PCG / pattern)
    profiles/r<NN>_<config>_rocprofv3_summary.txt  scripts/gpu_profile_config.sh <config> (FETCH_SIZE / WRITE_SIZE per kernel: measured bytes)

    python scripts/gen_design_table.py [--round 06] [--check]

writes the block between the markers `<!-- BEGIN GENERATED TABLE -->` / `<!-- END GENERATED TABLE -->` of DESIGN.md; --check only compares."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- BEGIN GENERATED TABLE -->", "<!-- END GENERATED TABLE -->"
ORDER = ["ns", "c5", "ns-perturbed", "c2", "c3", "c4"]
TITLE = {"ns": "**ns (headline): Hex8 linear elasticity 216^3**", "c5": "c5 at N = 1: Hex8 linear elasticity 256^3",
         "ns-perturbed": "ns-perturbed: the headline mesh, every vertex moved +-0.1 h (no affine element)", "c2": "c2: Hex8 Poisson 128^3",
         "c3": "c3: Tet4 linear elasticity, BCC res 75, numbering permuted", "c4": "c4: Hex27 NeoHookean 50 x 50 x 80, 27 points"}
KERNELS = {"ns": ("k_affine_records", "k_affine_rows<"), "c5": ("k_affine_records", "k_affine_rows<"), "c2": ("k_affine_records", "k_affine_rows<"),
           "ns-perturbed": ("k_hex8_rows",), "c3": ("k_gather_rows_tet4",), "c4": ("k_hex27_dense_blocks", "k_rows_from_tri")}


def measured_bytes(rnd, cfg):
    """FETCH_SIZE x 2 + WRITE_SIZE (KiB, per dispatch) of the configuration's kernels from its rocprofv3 summary; None when there is none"""
    path = os.path.join(ROOT, "profiles", f"r{rnd}_{cfg}_rocprofv3_summary.txt")
    if not os.path.exists(path):
        return None
    fetch = write = 0.0
    seen = False
    for ln in open(path):
        m = re.search(r"(k_[A-Za-z0-9_]+)\S*\s+.*?(FETCH_SIZE|WRITE_SIZE)\s+per-dispatch\s+([0-9.e+]+)", ln)
        if not m or not any(m.group(1).startswith(k.rstrip("<")) for k in KERNELS[cfg]):
            continue
        seen = True
        if m.group(2) == "FETCH_SIZE":
            fetch += float(m.group(3))
        else:
            write += float(m.group(3))
    return (2.0 * fetch + write) * 1024.0 if seen else None


def fmt_e(x):
    return f"{x:.3g}".replace("e+0", "e").replace("e+", "e")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="06")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    rnd = args.round
    d = json.load(open(os.path.join(ROOT, "profiles", f"r{rnd}_bench_default_n1.json")))
    rows = []
    head = ("| configuration (`python bench.py`, one MI355X) | kernel(s) | ms per assembly | elements/s | roofline: achieved of peak | fraction | "
            "measured HBM bytes / algorithmic | first assembly (tables) |")
    rows.append(head)
    rows.append("|---|---|---|---|---|---|---|---|")
    r = d["roofline"]
    ab = r["algorithmic_bytes_per_launch"]
    traffic = r.get("traffic")
    rows.append(f"| {TITLE['ns']} | `{'` + `'.join(KERNELS['ns']).replace('<', '')}` | **{d['ms_per_step']:.3f}** (first placement "
                f"{d['config'].get('first_placement_ms', float('nan')):.3f}) | {fmt_e(d['value'])} | {r['achieved']:.0f} of {r['peak']:.0f} {r['unit']} | "
                f"**{r['frac']:.3f}** (first placement {d['config'].get('first_placement_frac', float('nan')):.3f}) | "
                f"{(traffic / ab):.2f} ({traffic / 1e9:.2f} of {ab / 1e9:.2f} GB)" + f" | {d['config']['first_assembly_s'] * 1e3:.0f} ms |")
    for cfg in ORDER[1:]:
        s = d.get("secondary", {}).get(cfg)
        if not s:
            continue
        mb = measured_bytes(rnd, cfg)
        alg = s.get("algorithmic_bytes")
        ratio = f"{mb / alg:.2f} ({mb / 1e9:.2f} of {alg / 1e9:.2f} GB)" if (mb and alg) else "-"
        unit = s["unit"]
        ach = f"{s['achieved']:.0f}" if unit == "GB/s" else f"{s['achieved']:.1f}"
        peak = f"{s['peak']:.0f}" if unit == "GB/s" else f"{s['peak']:.1f}"
        kern = "`" + "` + `".join(k.rstrip("<") for k in KERNELS[cfg]) + "`"
        rows.append(f"| {TITLE[cfg]} | {kern} | {s['ms']:.3f} | {fmt_e(s['elements_per_s'])} | {ach} of {peak} {unit}" + (" (fp64 matrix cores)" if s["bound"] == "mfma" else "") +
                    f" | {s['frac']:.3f} | {ratio} | {s['first_assembly_s'] * 1e3:.0f} ms |")
    sv = d.get("secondary_vectors")
    if sv:
        rows.append(f"| residual f(u) of the headline mesh and operator | `{sv['residual_kernel']}` | {sv['residual_ms']:.3f} | {fmt_e(d['config']['elements_per_gpu'] / sv['residual_ms'] * 1e3)} | - | - | - | - |")
        rows.append(f"| energy (assemble_scalar), same | `{sv['energy_kernel']}` | {sv['energy_ms_blocking_call']:.3f} (blocking call) | - | - | - | - | - |")
    other = os.path.join(ROOT, "profiles", f"r{rnd}_other_kernels.jsonl")
    if os.path.exists(other):
        for ln in open(other):
            ln = ln.strip()
            if not ln.startswith("{"):
                continue
            o = json.loads(ln)
            if o.get("one_time") or "ms" not in o:
                pass
            name = o.get("kernel", "?")
            if any(t in name for t in ("SpMV", "PCG", "source vector", "NeoHookean", "StVK", "pattern build")):
                gbs = o["algorithmic_bytes"] / (o["ms"] * 1e-3) / 1e9 if o.get("algorithmic_bytes") else None
                rows.append(f"| {name.split(' (')[0]} ({o.get('config', '')}) | `{name.split('(')[-1].rstrip(')')}` | {o['ms']:.3f} | - | "
                            + (f"{gbs:.0f} of 8000 GB/s | {gbs / 8000.0:.3f}" if gbs else "- | -") + " | - | - |")
    cb = d.get("cpu_baseline")
    if cb:
        rows.append("")
        rows.append(f"CPU path beside it (`cpu_baseline`, kind `{cb['kind']}`): {fmt_e(cb['value'])} elements/s on {cb['cores']} threads "
                    f"(serial {fmt_e(cb.get('serial_value', 0.0))}); sample: {cb['sample']}")
    block = BEGIN + "\n" + f"(generated by `scripts/gen_design_table.py --round {rnd}` from `profiles/r{rnd}_bench_default_n1.json`, `profiles/r{rnd}_other_kernels.jsonl` and the "
    block += f"`profiles/r{rnd}_<config>_rocprofv3_summary.txt` files; do not edit by hand)\n\n" + "\n".join(rows) + "\n" + END
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    if BEGIN not in text or END not in text:
        sys.exit("DESIGN.md has no generated-table markers")
    new = text[:text.index(BEGIN)] + block + text[text.index(END) + len(END):]
    if args.check:
        if new != text:
            sys.exit("DESIGN.md section 3.4 table is not what the profiles give: run scripts/gen_design_table.py")
        print("DESIGN.md table is current")
        return
    open(path, "w").write(new)
    print(f"DESIGN.md: table regenerated from round {rnd} profiles ({len(rows)} lines)")


if __name__ == "__main__":
    main()
