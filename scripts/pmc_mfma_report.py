#!/usr/bin/env python3
"""Per-kernel averages of the PMC passes of scripts/gpu_round1al.sh (MFMA utilisation, LDS bank conflicts).
Usage: pmc_mfma_report.py <dir> [<dir> ...]"""
import csv
import glob
import json
import sys


def main():
    out = {}
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if not any(k in name for k in ("k_gram_mfma", "k_blockmul_mfma", "k_spmm", "k_spmv_sell")):
                    continue
                short = name.split("(")[0].replace("void ", "")
                out.setdefault(short, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    rep = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())} for k, cs in out.items()}
    for k, cs in rep.items():
        if "SQ_LDS_BANK_CONFLICT" in cs and cs.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            cs["lds_bank_conflict_fraction_of_active_cycles"] = cs["SQ_LDS_BANK_CONFLICT"] / cs["SQ_LDS_IDX_ACTIVE"]
    print(json.dumps(rep, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
