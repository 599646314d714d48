"""Summarise one rocprofv3 --pmc pass of SQ counters over tools/traffic_probe.py into per-kernel wave-cycle shares.

    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
              SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d out -- python3 tools/traffic_probe.py
    python tools/pmc_sq_report.py out profiles/rNN_sq_counters_Cstar.json

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); WAIT_ANY (parked on s_waitcnt /
barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES."""
import collections
import csv
import glob
import json
import sys


def main(d, out):
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        if "psvo::" not in k or "adam" in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    rep = {}
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        n = max(1, len(disp[k]))
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        rep[k[:90]] = {
            "launches": n,
            "waves_per_launch": c.get("SQ_WAVES", 0) / n,
            "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1),
            "lds_insts_per_wave": c.get("SQ_INSTS_LDS", 0) / max(c.get("SQ_WAVES", 1), 1),
            "wave_cycles_per_wave": 4.0 * wc / max(c.get("SQ_WAVES", 1), 1),
            "share_active_valu": c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            "share_active_any": c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            "share_wait_any": c.get("SQ_WAIT_ANY", 0) / wc,
            "share_wait_inst_any": c.get("SQ_WAIT_INST_ANY", 0) / wc,
        }
    json.dump(rep, open(out, "w"), indent=1)
    for k, v in rep.items():
        print("%-70s waves %7.0f  VALU/wave %8.0f  cyc/wave %9.0f  active_valu %.2f  wait_any %.2f  wait_inst %.2f"
              % (k[:70], v["waves_per_launch"], v["valu_insts_per_wave"], v["wave_cycles_per_wave"],
                 v["share_active_valu"], v["share_wait_any"], v["share_wait_inst_any"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
