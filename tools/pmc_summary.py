"""Turns the two counter passes of tools/traffic_probe.py into profiles/hbm_traffic_by_entry.json (+ .csv).

    python tools/pmc_summary.py <fetch_dir> <write_dir> <manifest.json> <round tag> [<sq_dir>]
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; gfx950 tallies 128-byte read requests at 64 bytes).
With <sq_dir> (a pass with SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
GRBM_GUI_ACTIVE) also profiles/<tag>_mfma_busy_by_entry.csv and profiles/mfma_busy_by_entry.json: per entry point,
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)  (busy cycles are summed over
the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs), valu_busy_frac likewise from SQ_ACTIVE_INST_VALU (quad-cycles x 4).
"""
import csv, glob, json, sys


def segments(d, counter, marker_grid):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    segs, cur = [], None
    for r in rows:
        if "FillFunctor" in r["Kernel_Name"] and int(r["Grid_Size"]) == marker_grid:
            cur = []
            segs.append(cur)
        elif cur is not None:
            cur.append((r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["Counter_Value"]) * 1024.0))
    return segs


def sq_summary(sq_dir, man, marker_grid, tag):
    f = glob.glob(sq_dir + "/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    by_disp = {}
    for r in rows:
        d = by_disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    segs, cur = [], None
    for did in sorted(by_disp):
        d = by_disp[did]
        if "FillFunctor" in d["name"] and d["grid"] == marker_grid:
            cur = []
            segs.append(cur)
        elif cur is not None:
            cur.append(d)
    out, lines = {"_note": f"matrix-core / vector-unit busy fractions per entry point at the cfg3 shapes; {tag}; rocprofv3 --pmc "
                           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE over "
                           "tools/traffic_probe.py; mfma_busy_frac = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)"}, \
        ["entry,kernel,mfma_busy_cycles,gui_active,mfma_busy_frac,valu_busy_frac,wait_any_frac"]
    for name, seg in zip(man["segments"], segs):
        if name in ("end", "setup"):
            continue
        for d in seg:
            k = d["name"].split("(")[0].replace("void ", "")
            if k.startswith("at::") or "rocclr" in k or d.get("GRBM_GUI_ACTIVE", 0) <= 0:
                continue
            simd_cycles = d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
            va = 4.0 * d.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles
            wa = d.get("SQ_WAIT_ANY", 0.0) / max(d.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            key = f"{name.replace(':E', '')}|{k[:64]}"
            out[key] = {"mfma_busy_frac": round(mf, 4), "valu_busy_frac": round(va, 4), "wait_any_frac": round(wa, 4),
                        "gui_active": d["GRBM_GUI_ACTIVE"]}
            lines.append(f"{name},{k[:60]},{d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f},{d['GRBM_GUI_ACTIVE']:.0f},{mf:.4f},{va:.4f},{wa:.4f}")
    json.dump(out, open("profiles/mfma_busy_by_entry.json", "w"), indent=1)
    open(f"profiles/{tag}_mfma_busy_by_entry.csv", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def main():
    fetch_dir, write_dir, manifest, tag = sys.argv[1:5]
    man = json.load(open(manifest))
    # torch's vectorised fill launches numel/4 threads rounded up to 1024-thread groups... find the grid empirically
    f = glob.glob(fetch_dir + "/**/*counter_collection.csv", recursive=True)[0]
    grids = {}
    for r in csv.DictReader(open(f)):
        if "FillFunctor" in r["Kernel_Name"]:
            grids[int(r["Grid_Size"])] = grids.get(int(r["Grid_Size"]), 0) + 1
    marker_grid = max((g for g, c in grids.items() if c == len(man["segments"])), default=None)
    if marker_grid is None:
        raise SystemExit(f"marker fill not found; fill grids seen: {grids}")
    F, W = segments(fetch_dir, "FETCH_SIZE", marker_grid), segments(write_dir, "WRITE_SIZE", marker_grid)
    out = {"_note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes, KiB counters; "
                    f"gfx950 read correction); cfg3 shapes; {tag}; made by tools/traffic_probe.py + tools/pmc_summary.py"}
    lines = ["entry,kernels,read_bytes,write_bytes,total_bytes"]
    for name, fs, ws in zip(man["segments"], F, W):
        if name in ("end", "setup"):
            continue
        big = [(k, v) for k, v in fs if v > 0]
        rd = 2.0 * sum(v for _, v in fs)
        wr = sum(v for _, v in ws)
        kern = "+".join(sorted({k for k, _ in fs if not k.startswith("at::") and "rocclr" not in k}))
        if "+bwd_weight" in name:            # split the autograd backward into its two GEMM calls by kernel family
            shape = name.split(":")[1]
            def part(pred):
                r_ = 2.0 * sum(v for k, v in fs if pred(k)); w_ = sum(v for k, v in ws if pred(k))
                return r_, w_
            rw, ww = part(lambda k: "tswgrad" in k)
            ri, wi = part(lambda k: "tsgemm" in k)
            if rw and ri:
                out[f"mbvae_linear_bwd_weight:{shape}"] = {"bytes": round(rw + ww), "read": round(rw), "write": round(ww)}
                out[f"mbvae_linear_bwd_input:{shape}"] = {"bytes": round(ri + wi), "read": round(ri), "write": round(wi)}
                lines.append(f"mbvae_linear_bwd_weight:{shape},tswgrad,{rw:.0f},{ww:.0f},{rw + ww:.0f}")
                lines.append(f"mbvae_linear_bwd_input:{shape},tsgemm2,{ri:.0f},{wi:.0f},{ri + wi:.0f}")
                continue
        key = name.replace(":E", "")
        out[key] = {"bytes": round(rd + wr), "read": round(rd), "write": round(wr)}
        lines.append(f"{name},{kern[:80]},{rd:.0f},{wr:.0f},{rd + wr:.0f}")
    json.dump(out, open("profiles/hbm_traffic_by_entry.json", "w"), indent=1)
    open(f"profiles/{tag}_hbm_traffic_by_entry.csv", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if len(sys.argv) > 5:
        sq_summary(sys.argv[5], man, marker_grid, tag)


if __name__ == "__main__":
    main()
