#!/usr/bin/env python3
"""Summarise rocprofv3 outputs (kernel stats + PMC counter CSVs) into profiles/<tag>_*.{md,json}."""
import csv, glob, json, os, sys, collections

def load_counters(d):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if any(k in name for k in ("gemm", "attn", "prefill32", "layernorm", "patch_embed", "rope", "gather_rows")):
                key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                if ("prefill32" in key or "attn32" in key) and r.get("Grid_Size"):
                    key += f" [grid {r['Grid_Size']}]"       # one row per launch shape (8 x 1216 and 1 x 9280; cut and uncut)
                rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                rows[key]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                rows[key]["_vgpr"].append(float(r["VGPR_Count"]) + float(r.get("Accum_VGPR_Count", 0) or 0))
                rows[key]["_lds"].append(float(r["LDS_Block_Size"]))
    return rows

def main():
    root, tag = sys.argv[1], sys.argv[2]
    out = {}
    for sub in ("prof_pmc_sq", "prof_pmc_fetch", "prof_pmc_write"):
        rows = load_counters(os.path.join(root, sub))
        for k, c in rows.items():
            o = out.setdefault(k, {})
            for name, vals in c.items():
                if name.startswith("_"):
                    o.setdefault(name[1:], round(sum(vals) / len(vals), 1))
                else:
                    o[name] = round(sum(vals) / len(vals), 1)     # mean per dispatch
    for k, o in out.items():
        if "SQ_BUSY_CYCLES" in o and "SQ_VALU_MFMA_BUSY_CYCLES" in o and o["SQ_BUSY_CYCLES"]:
            o["mfma_busy_over_sq_busy"] = round(o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["SQ_BUSY_CYCLES"], 4)
        if "SQ_LDS_IDX_ACTIVE" in o and o.get("SQ_LDS_IDX_ACTIVE"):
            o["lds_bank_conflict_frac"] = round(o.get("SQ_LDS_BANK_CONFLICT", 0) / o["SQ_LDS_IDX_ACTIVE"], 4)
        if "FETCH_SIZE" in o:
            # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request on wide streams -> x2
            # (MI355X_MICROARCH.md section HBM)
            o["hbm_read_bytes_corrected"] = round(o["FETCH_SIZE"] * 1024 * 2)
        if "WRITE_SIZE" in o:
            o["hbm_write_bytes"] = round(o["WRITE_SIZE"] * 1024)
    os.makedirs("profiles", exist_ok=True)
    # which build the passes ran on (the GPU box has no .git: the run script hands the commit over in SLIME_GIT_HEAD); bench.py prints
    # it as roofline.traffic_head next to the traffic figure it reads from this file
    # ... and a digest of the kernel sources the profiled library was built from (slime_amd/csrc): bench.py recomputes it and says in
    # `roofline.traffic_current` whether the committed passes describe the build that is running (the r05 figure predated
    # later staging changes and only the commit stamp said so)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from slime_amd._lib import csrc_digest
    out["_meta"] = {"git_head": os.environ.get("SLIME_GIT_HEAD", "unknown"), "csrc_sha": csrc_digest(), "target": "tools/pmc_target.py",
                    "passes": ["prof_pmc_sq", "prof_pmc_fetch", "prof_pmc_write"]}
    json.dump(out, open(f"profiles/{tag}_pmc_kernels.json", "w"), indent=1, sort_keys=True)
    del out["_meta"]
    # derived figures per kernel: MFMA-busy fraction of the chip's SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over
    # SIMDs; GRBM_GUI_ACTIVE x 128 matches the sum at 100 % on this part), LDS conflict share, where the wave cycles went
    summ = {}
    for k, o in out.items():
        r = {"dur_us": round(o.get("dur_ns", 0) / 1e3, 1)}
        if "hbm_read_bytes_corrected" in o: r["fabric_read_MB_x2corrected"] = round(o["hbm_read_bytes_corrected"] / 1e6, 1)
        if "hbm_write_bytes" in o: r["fabric_write_MB"] = round(o["hbm_write_bytes"] / 1e6, 1)
        r["lds_bank_conflict_frac"] = o.get("lds_bank_conflict_frac")
        if o.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in o:
            r["mfma_busy_frac"] = round(o["SQ_VALU_MFMA_BUSY_CYCLES"] / (o["GRBM_GUI_ACTIVE"] * 128.0), 3)
        if o.get("SQ_WAVE_CYCLES"):
            w = o["SQ_WAVE_CYCLES"]
            r["wave_cycles_split"] = {"parked_waitcnt_barrier": round(o.get("SQ_WAIT_ANY", 0) / w, 3),
                                      "issue_stalled": round(o.get("SQ_WAIT_INST_ANY", 0) / w, 3),
                                      "issuing": round(o.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)}
        summ[k] = r
    json.dump(summ, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
    print(json.dumps(summ, indent=1, sort_keys=True))

if __name__ == "__main__":
    main()
