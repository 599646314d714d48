"""Combine the FETCH_SIZE and WRITE_SIZE passes of tools/traffic_probe.py into HBM bytes per launch.

Calibration (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are in KiB-like units derived from
64-byte request counts and under-count some access widths on gfx950, so both are scaled by the
factor that makes the calibration launch (adam on n floats: 16 n bytes read, 12 n bytes written,
4 bytes per lane like the PSVO kernels) come out right."""
import collections
import csv
import glob
import json
import sys

n = 32 * 1024 * 1024


def load(d):
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])))
    out = collections.defaultdict(list)
    for r in rows:
        out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main(fdir, wdir, outpath):
    F, W = load(fdir), load(wdir)
    cal_name = [k for k in F if "adam_kernel" in k][0]
    f_cal, w_cal = max(F[cal_name]), max(W[cal_name])       # the 32 Mi-element launch
    kf = 16.0 * n / f_cal                                  # bytes per counter unit, reads
    kw = 12.0 * n / w_cal                                  # ... writes
    rep = {"calibration": {"kernel": "psvo::adam_kernel on %d floats" % n, "read_bytes": 16 * n, "write_bytes": 12 * n,
                           "FETCH_SIZE": f_cal, "WRITE_SIZE": w_cal, "bytes_per_FETCH_unit": kf,
                           "bytes_per_WRITE_unit": kw}, "kernels": {}}
    for k in sorted(F):
        if "psvo::" not in k or "adam" in k:
            continue
        f = sum(F[k]) / len(F[k])
        w = sum(W.get(k, [0])) / max(1, len(W.get(k, [0])))
        rep["kernels"][k[:90]] = {"launches": len(F[k]), "read_MB_per_launch": f * kf / 1e6,
                                  "write_MB_per_launch": w * kw / 1e6, "hbm_MB_per_launch": (f * kf + w * kw) / 1e6}
    json.dump(rep, open(outpath, "w"), indent=1)
    for k, v in rep["kernels"].items():
        print("%-90s %8.1f MB read %8.1f MB written" % (k, v["read_MB_per_launch"], v["write_MB_per_launch"]))
    print("calibration: %.1f B/unit read, %.1f B/unit write" % (kf, kw))


if __name__ == "__main__":
    main(*sys.argv[1:4])
