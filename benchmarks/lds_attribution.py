"""Table of the LDS bank-conflict attribution passes of benchmarks/lds_attribution.sh:
    python3 benchmarks/lds_attribution.py gpurun_out/<tag>
For every workload: the trace kernel's LDS counters per launch with the product build, and for every RT_EXP build
(one kind of LDS access issued twice) the growth of the counters over it -- that access kind's LDS-array cycles,
the conflict cycles among them and its conflict rate."""
import csv
import glob
import json
import os
import sys

KINDS = {
    "51": "ray table read by owner, pair filter (3 ds_read_b128 per pass)",
    "52": "pair image read by group, pair filter (4 or 8 ds_read_b128 per pass; scenes resident in LDS)",
    "53": "ray table read by owner, exact evaluation (2 ds_read_b128 + 1 ds_read_b64 per pass)",
    "54": "64-bit minimum on best[owner] (ds_min_u64)",
    "55": "(lane, group) list read (ds_read_u16) and group size read (ds_read_u8)",
    "56": "(lane, group) list write (ds_write_b16)",
    "57": "(lane, sphere) list write and read (ds_write_b32 / ds_read_b32, consecutive addresses)",
    "58": "per-lane ray table and key write (3 ds_write_b128 + 1 ds_write_b64 per trip)",
}


def counters(d):
    sums, calls = {}, set()
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "trace_kernel" not in row["Kernel_Name"]:
                    continue
                sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                calls.add(row["Dispatch_Id"])
    n = max(1, len(calls))
    return {k: v / n for k, v in sums.items()}


root = sys.argv[1]
out = {}
for wdir in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(os.path.join(wdir, "base")):
        continue
    base = counters(os.path.join(wdir, "base"))
    if not base:
        continue
    w = {"product_build_per_launch": base,
         "conflict_rate": base["SQ_LDS_BANK_CONFLICT"] / base["SQ_LDS_IDX_ACTIVE"], "by_access_kind": []}
    explained_conf = explained_act = 0.0
    for k, text in KINDS.items():
        c = counters(os.path.join(wdir, k))
        if not c:
            continue
        d_conf = c["SQ_LDS_BANK_CONFLICT"] - base["SQ_LDS_BANK_CONFLICT"]
        d_act = c["SQ_LDS_IDX_ACTIVE"] - base["SQ_LDS_IDX_ACTIVE"]
        d_inst = c["SQ_INSTS_LDS"] - base["SQ_INSTS_LDS"]
        explained_conf += d_conf
        explained_act += d_act
        w["by_access_kind"].append({"exp": int(k), "access": text, "lds_instructions": d_inst, "lds_array_cycles": d_act, "conflict_cycles": d_conf,
                                    "conflict_rate": d_conf / d_act if d_act > 0 else None,
                                    "share_of_all_conflict_cycles": d_conf / base["SQ_LDS_BANK_CONFLICT"],
                                    "share_of_all_lds_cycles": d_act / base["SQ_LDS_IDX_ACTIVE"],
                                    "array_cycles_per_instruction": d_act / d_inst if d_inst > 0 else None})
    w["conflict_cycles_explained"] = explained_conf / base["SQ_LDS_BANK_CONFLICT"]
    w["lds_cycles_explained"] = explained_act / base["SQ_LDS_IDX_ACTIVE"]
    out[os.path.basename(wdir)] = w
print(json.dumps(out, indent=1))
