#!/usr/bin/env python3
"""Combine the FETCH_SIZE and WRITE_SIZE passes of scripts/pmc_spmv.py into per-launch HBM traffic.
Usage: pmc_report.py <fetch_dir> <write_dir> <algorithmic_bytes> <calib_elems>"""
import csv
import glob
import json
import sys


def load(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def main():
    fd, wd, alg, calib = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    rep = {}
    kax = [k for k in fetch if "k_axpby" in k]
    # FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B (guide: hbm_bytes = (FETCH+WRITE)*1024);
    # on gfx950 FETCH_SIZE under-reports wide streams: calibrate on the known 16 B/element read of axpby
    corr_r = corr_w = 1.0
    if kax:
        fr = sum(fetch[kax[0]]) / len(fetch[kax[0]]) * 1024.0
        corr_r = (16.0 * calib) / fr
        rep["calibration"] = {"kernel": "k_axpby", "expected_read_bytes": 16.0 * calib, "FETCH_SIZE_bytes": fr,
                              "read_correction": corr_r}
        kw = [k for k in write if "k_axpby" in k]
        if kw:
            wr = sum(write[kw[0]]) / len(write[kw[0]]) * 1024.0
            corr_w = (8.0 * calib) / wr
            rep["calibration"].update({"expected_write_bytes": 8.0 * calib, "WRITE_SIZE_bytes": wr,
                                       "write_correction": corr_w})
    for name in ("k_spmv_sell", "k_spmv_lds"):
        kf = [k for k in fetch if name in k]
        kw = [k for k in write if name in k]
        if not kf:
            continue
        fr = sum(fetch[kf[0]]) / len(fetch[kf[0]]) * 1024.0
        wr = sum(write[kw[0]]) / len(write[kw[0]]) * 1024.0 if kw else 0.0
        rep[name] = {"launches": len(fetch[kf[0]]), "FETCH_SIZE_bytes_raw": fr, "WRITE_SIZE_bytes_raw": wr,
                     "traffic_bytes_corrected": fr * corr_r + wr * corr_w, "algorithmic_bytes": alg,
                     "traffic_over_algorithmic": (fr * corr_r + wr * corr_w) / alg}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
