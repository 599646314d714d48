#!/usr/bin/env python3
"""Kernel-by-kernel timeline of ONE replayed training step from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline
    python tools/step_timeline.py DIR [out.txt]
Prints every dispatch of the last training step (from the previous step's Adam launch to its own): start (us, relative), duration, queue, name."""
import csv
import glob
import sys


def main(src, dst=None):
    rows = []
    for path in glob.glob(src + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[3]]
    if len(marks) < 2:
        raise SystemExit("fewer than two training steps in the trace")
    # a training step = everything after the previous step's Adam launch up to and including its own
    lo, hi = marks[-2] + 1, marks[-1] + 1
    t0 = rows[lo][0]
    out = []
    for s, e, q, name in rows[lo:hi]:
        short = name.split("(")[0].replace("void ", "").replace("psvo::", "")
        if "at::native" in short:
            short = "torch:" + short.split("at::native::")[-1][:60]
        out.append("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short[:90]))
    text = "\n".join(["# start_us   dur_us  queue kernel   (one replayed C* training step, first launch = 0)"] + out)
    print(text)
    if dst:
        open(dst, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
