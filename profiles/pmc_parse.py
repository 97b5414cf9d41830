"""Summarise the counter_collection CSVs of profiles/pmc_collect.sh into one JSON record per workload and kernel."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

# field_fwd_kernel<T, MT, NT, WAVES = 8, KEEP, DIR>: KEEP (the stash-writing training forward) follows the 8
NAMES = [("field_fwd_kernel", ("Li8ELb1", "8, true"), "field_fwd_full"), ("field_fwd_kernel", ("Li8ELb0", "8, false"), "field_fwd_sigma"),
         ("field_bwd_kernel", (), "field_bwd_chain"), ("wgrad256_kernel", (), "wgrad"), ("skinny_wgrad_kernel", (), "skinny_wgrad"),
         ("field_adjoint_kernel", (), "field_adjoint"), ("field_adjbwd_kernel", (), "field_adjoint_bwd"),
         # round 3: the merged-set compositing kernel (MODE 0 forward, 1 Lambertian tail, 2 backward), pass-1 compositing fused with the
         # guided resampling, the ray-level shading + loss kernel, multi-group Adam
         ("merged_composite_kernel", ("ILi0E", "<0,"), "composite_fwd"), ("merged_composite_kernel", ("ILi1E", "<1,", "ILi2E", "<2,"), "composite_bwd"),
         ("composite_guided_kernel", (), "guided_samples"), ("ray_shade_loss_kernel", (), "brdf"), ("adam_multi_kernel", (), "adam"),
         ("composite_kernel", ("Lb0", "<false"), "composite_fwd"), ("composite_kernel", ("Lb1", "<true"), "composite_bwd"),
         ("guided_kernel", (), "guided_samples"), ("adam_kernel", (), "adam")]


def classify(kernel_name):
    for sub, tags, out in NAMES:
        if sub in kernel_name and (not tags or any(t in kernel_name for t in tags)):
            return out
    return None


def main():
    tmp, out = sys.argv[1], sys.argv[2]
    rec = {"source_hash": bench.source_hash(), "workloads": {},
           "how": "rocprofv3 --pmc <group> --kernel-trace over profiles/prof_step.py 3 <config> <dtype>, one group per pass; per-launch means. "
                  "hbm_bytes = 2 x FETCH_SIZE(KB) x 1024 + WRITE_SIZE(KB) x 1024 (MI355X_MICROARCH.md section HBM: FETCH_SIZE tallies the "
                  "128-B requests of wide coalesced reads at 64 B on gfx950; WRITE_SIZE is exact for 16-B/lane stores and float atomics); "
                  "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)"}
    for wl in sys.argv[3:]:
        agg = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(tmp, f"pmcc_{wl}_*", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = classify(row["Kernel_Name"])
                if k is None:
                    continue
                a = agg[(k, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
        kern = {}
        for (k, c), (v, n) in agg.items():
            kern.setdefault(k, {})[c] = v / n
        for k, v in kern.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                v["read_bytes"] = 2 * v["FETCH_SIZE"] * 1024
                v["write_bytes"] = v["WRITE_SIZE"] * 1024
                v["hbm_bytes"] = v["read_bytes"] + v["write_bytes"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
                v["mfma_busy"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)
            if v.get("SQ_LDS_IDX_ACTIVE"):
                v["lds_conflict_share"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
            if "TCC_HIT_sum" in v:
                v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        rec["workloads"][wl] = kern
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
    for wl, kern in rec["workloads"].items():
        for k, v in sorted(kern.items()):
            print(wl, k, {kk: (round(vv, 4) if vv < 10 else int(vv)) for kk, vv in v.items() if kk in ("hbm_bytes", "read_bytes", "write_bytes", "mfma_busy", "l2_hit_rate", "lds_conflict_share")})


if __name__ == "__main__":
    main()
