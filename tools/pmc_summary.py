"""Turns the two counter passes of tools/traffic_probe.py into profiles/hbm_traffic_by_entry.json (+ .csv).

    python tools/pmc_summary.py <fetch_dir> <write_dir> <manifest.json> <round tag>
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; gfx950 tallies 128-byte read requests at 64 bytes).
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


if __name__ == "__main__":
    main()
