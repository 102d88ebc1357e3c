#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes over tools/conv_traffic_target.py (one counter set per pass, as
MI355X_MICROARCH.md section HBM prescribes) into the committed evidence:

    conv_pmc_summary.py <dir with the passes' *counter_collection.csv> <layers.json> <tag>
        -> profiles/<tag>_conv_hbm_traffic_by_layer.txt   (FETCH_SIZE / WRITE_SIZE vs algorithmic bytes per layer)
        -> profiles/<tag>_conv_mfma_util.md               (SQ_* issue counters per conv kernel)
        -> profiles/pmc_summary.json["conv_split_fast_kernel"]  (what bench.py reports as roofline.traffic)

Corrections (the guide's): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a
wide (16 B / lane) coalesced read stream -- the conv kernels' operand DMA is of that kind -- so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores (the epilogue's)."""
import collections
import csv
import glob
import json
import os
import sys

d, layers_json, tag = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = collections.defaultdict(dict)          # dispatch id -> {counter: value, "name":..., "grid":...}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    pass_id = os.path.dirname(f)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "conv_split" not in name and "conv_win32" not in name and "conv_poolin" not in name and "conv_regw32" not in name:
            continue
        key = (pass_id, int(r["Dispatch_Id"]))
        rows[key]["name"] = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        rows[key][r["Counter_Name"]] = rows[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# per pass: conv launches in dispatch order; the target runs 3 forwards -> keep the last third
by_pass = collections.defaultdict(list)
for (pass_id, disp), v in sorted(rows.items()):
    by_pass[pass_id].append(v)
layers = json.load(open(layers_json))["per_launch"]
nl = len(layers)
per_counter = {}                                # counter -> list over the nl launches of the last forward
names = None
for pass_id, lst in by_pass.items():
    assert len(lst) % nl == 0, (pass_id, len(lst), nl)
    last = lst[-nl:]
    names = [v["name"] for v in last]
    for c in last[0]:
        if c != "name":
            per_counter[c] = [v.get(c, 0.0) for v in last]
out_dir = os.path.join(ROOT, "profiles")
# ---- traffic --------------------------------------------------------------------------------------------------
if "FETCH_SIZE" in per_counter and "WRITE_SIZE" in per_counter:
    agg = collections.OrderedDict()
    for i, (name, ib, ob, wb) in enumerate(layers):
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ib; a[2] += per_counter["FETCH_SIZE"][i] * 1024 * 2; a[3] += ob
        a[4] += per_counter["WRITE_SIZE"][i] * 1024; a[5] += wb
    lines = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/conv_traffic_target.py: the conv launches of one",
             "# trunk forward at batch 500.  fetch = FETCH_SIZE KiB x 1024 x 2 (gfx950: wide coalesced reads are tallied at half),",
             "# write = WRITE_SIZE KiB x 1024.  Fabric-side counters: Infinity-Cache hits are included.  in / out = algorithmic",
             "# activation bytes (4 B per element: two fp16 planes); weights are added to the read side of the ratio.",
             f"{'layer':38s} {'n':>2s} {'in MB':>8s} {'fetch MB':>9s} {'ratio':>6s} {'out MB':>8s} {'write MB':>9s} {'ratio':>6s}"]
    tin = tf = tout = tw = twb = 0.0
    for name, (n, ib, fb, ob, wrb, wb) in sorted(agg.items(), key=lambda kv: -(kv[1][2] + kv[1][4])):
        lines.append(f"{name:38s} {n:2d} {ib / 1e6:8.0f} {fb / 1e6:9.0f} {fb / (ib + wb):6.2f} {ob / 1e6:8.0f} {wrb / 1e6:9.0f} {wrb / ob:6.2f}")
        tin += ib; tf += fb; tout += ob; tw += wrb; twb += wb
    lines.append(f"{'total':38s} {nl:2d} {tin / 1e6:8.0f} {tf / 1e6:9.0f} {tf / (tin + twb):6.2f} {tout / 1e6:8.0f} {tw / 1e6:9.0f} {tw / tout:6.2f}")
    alg = tin + tout + twb
    lines.append(f"# all conv launches: {(tf + tw) / 1e9:.1f} GB moved vs {alg / 1e9:.1f} GB algorithmic = {(tf + tw) / alg:.2f}x")
    open(os.path.join(out_dir, f"{tag}_conv_hbm_traffic_by_layer.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    pj_path = os.path.join(out_dir, "pmc_summary.json")
    pj = json.load(open(pj_path)) if os.path.exists(pj_path) else {}
    pj["conv_split_fast_kernel"] = {
        "launches_per_forward": nl, "fetch_correction": 2.0, "hbm_bytes_per_forward": tf + tw,
        "hbm_bytes_per_launch": (tf + tw) / nl, "algorithmic_bytes_per_forward": alg,
        "algorithmic_bytes_per_launch": alg / nl, "ratio": (tf + tw) / alg,
        "kernels": sorted(set(names)),
        "note": f"all {nl} conv launches of one trunk forward at batch 500; per layer: profiles/{tag}_conv_hbm_traffic_by_layer.txt; "
                "fabric-side counters (Infinity-Cache hits included)"}
    pj["_source"] = f"{tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/conv_traffic_target.py, condensed by tools/conv_pmc_summary.py"
    json.dump(pj, open(pj_path, "w"), indent=1)
# ---- issue counters -------------------------------------------------------------------------------------------
want = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES",
        "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_ACTIVE_INST_LDS"]
have = [c for c in want if c in per_counter]
if have:
    agg = collections.OrderedDict()
    for i, name in enumerate(names):
        a = agg.setdefault(name, collections.defaultdict(float))
        a["launches"] += 1
        for c in have:
            a[c] += per_counter[c][i]
    tot = collections.defaultdict(float)
    for a in agg.values():
        for c, v in a.items():
            tot[c] += v
    agg["ALL CONV LAUNCHES"] = tot

    def ratio(a, x, y, scale=1.0):
        return f"{a[x] / a[y] * scale:.3f}" if x in a and y in a and a[y] else "n/a"
    md = [f"# {tag}: SQ issue counters of the convolution kernels", "",
          "rocprofv3 `--pmc` passes (one counter group per pass; program directly after `--`) over `tools/conv_traffic_target.py`:",
          "the conv launches of ONE trunk forward at batch 500 (the last of three), summed per kernel instance.", "",
          "* `MFMA util` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): share of ALL SIMD cycles of the launch in",
          "  which the SIMD's matrix core was executing (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs and the SQ counters over all",
          "  SIMDs; cross-check: SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA, the 4 passes x 4 cycles of a 16x16x32 MFMA)",
          "* `vs clock-free peak`: the same figure is fp16-MFMA rate / (1024 SIMDs x 1024 flop/clk x the clock the chip actually ran)",
          "* `VALU:MFMA` = SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU counts the MFMAs as well)",
          "* `issue-stall` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, `parked` = SQ_WAIT_ANY / SQ_WAVE_CYCLES", "",
          "* `LDS conflict` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent replaying bank conflicts / cycles it was",
          "  busy with indexed operations: ds_read / ds_write, not the LDS-DMA fills), `LDS:MFMA` = SQ_INSTS_LDS / SQ_INSTS_MFMA", "",
          "| kernel | launches | MFMA util | VALU:MFMA | issue-stall | parked | LDS conflict | LDS:MFMA | SQ_INSTS_MFMA |",
          "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for name, a in agg.items():
        md.append(f"| `{name}` | {int(a['launches'])} | {ratio(a, 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 8.0 / 1024.0)} | "
                  f"{ratio(a, 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA')} | {ratio(a, 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES')} | "
                  f"{ratio(a, 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')} | {ratio(a, 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')} | "
                  f"{ratio(a, 'SQ_INSTS_LDS', 'SQ_INSTS_MFMA')} | {a.get('SQ_INSTS_MFMA', 0):.3e} |")
    md += ["", "raw sums: " + ", ".join(f"{c} {tot[c]:.4e}" for c in have)]
    open(os.path.join(out_dir, f"{tag}_conv_mfma_util.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))
