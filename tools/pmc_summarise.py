"""Per-launch means of every counter rocprofv3 collected for ONE kernel (name substring) under a pmc_collect.sh output
directory -> JSON on stdout.  Counter values of a dispatch are summed over the rows rocprofv3 emits for it (one per XCD /
SE instance), then averaged over the dispatches of the kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, kern = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))       # counter -> dispatch -> sum
meta = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if kern not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
            meta = {"kernel_name": row["Kernel_Name"][:120], "VGPR": row["VGPR_Count"], "AGPR": row["Accum_VGPR_Count"],
                    "SGPR": row["SGPR_Count"], "LDS": row["LDS_Block_Size"], "scratch": row["Scratch_Size"],
                    "grid": row["Grid_Size"], "workgroup": row["Workgroup_Size"]}
out = dict(meta)
for name, per in sorted(acc.items()):
    out[name] = sum(per.values()) / max(1, len(per))
    out.setdefault("launches", len(per))
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:     # KB units; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md)
    out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0
if "SQ_WAVE_CYCLES" in out and "SQ_WAIT_ANY" in out:
    out["wait_any_frac"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
    out["wait_inst_any_frac"] = out.get("SQ_WAIT_INST_ANY", 0.0) / out["SQ_WAVE_CYCLES"]
if out.get("SQ_LDS_IDX_ACTIVE"):
    out["lds_bank_conflict_frac"] = out.get("SQ_LDS_BANK_CONFLICT", 0.0) / out["SQ_LDS_IDX_ACTIVE"]
print(json.dumps(out, indent=1))
