#!/usr/bin/env python3
"""A/B table of the reverse backward-simulation kernel variants (psvo_set_tuning, PSVO_TUNE_BSIM_BWD) at a bench workload:
kernel time from bench.py's HIP events, SQ counters from two rocprofv3 --pmc passes over tools/traffic_probe.py.

    python tools/ab_bsim_bwd.py run "C*" gpurun_out/ab          (GPU box: writes gpurun_out/ab/*.json / *.csv)
    python tools/ab_bsim_bwd.py report gpurun_out/ab profiles/r02_bsim_bwd_ab.md "C*"

Counters (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
cycles).  Every figure is per wave of the kernel (2048 waves at C*), divided by T for the per-time-step columns."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {
    "a": "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS",
    "b": "SQ_WAVES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES "
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32",
}
VARIANTS = (0, 1, 2, 3, 4)
NAMES = {0: "v1: lane = (chain, half, m), per-j butterflies", 1: "v2: j on lanes, VALU",
         2: "v2 + per-j sums on MFMA 16x16x4 f32", 3: "v2 + pair exponents on MFMA 16x16x4 f32",
         4: "v2 + pair exponents on MFMA 16x16x32 bf16 (3-piece split)"}


def run(workload, out):
    out = os.path.abspath(out)
    os.makedirs(out, exist_ok=True)
    for v in VARIANTS:
        env = dict(os.environ, PSVO_BSIM_BWD_VARIANT=str(v), PSVO_WORKLOAD=workload, TMPDIR="/tmp")
        with open(os.path.join(out, "bench_v%d.json" % v), "w") as f:
            subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "30",
                            "--warmup", "10", "--no-cpu-baseline", "--bsim-bwd-variant", str(v)], stdout=f,
                           stderr=subprocess.DEVNULL, env=env, cwd=ROOT, timeout=300)
        for tag, ctr in PASSES.items():
            d = os.path.join(out, "pmc_%s_v%d" % (tag, v))
            subprocess.run(["rocprofv3", "--pmc"] + ctr.split() + ["--output-format", "csv", "-d", d, "--",
                            "python3", os.path.join(ROOT, "tools", "traffic_probe.py")], env=env, cwd="/tmp",
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            rows = collections.defaultdict(float)
            n = set()
            for path in glob.glob(d + "/*/*_counter_collection.csv"):
                for r in csv.DictReader(open(path)):
                    if "bsim_bwd" in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"]:
                        rows[r["Counter_Name"]] += float(r["Counter_Value"])
                        n.add(r["Dispatch_Id"])
            rows["launches"] = len(n)
            json.dump(rows, open(os.path.join(out, "pmc_%s_v%d.json" % (tag, v)), "w"))
            subprocess.run(["rm", "-rf", d])
        print("variant", v, "done", flush=True)


def report(src, dst, workload):
    sys.path.insert(0, ROOT)
    import bench
    T = bench.WORKLOADS[workload][2]
    lines = ["# psvo_bsim_backward at %s: variants of the reverse backward-simulation kernel (round 2)" % workload, "",
             "`python tools/ab_bsim_bwd.py run \"%s\" gpurun_out/ab` on one MI355X; kernel time = median over 10 steps of HIP "
             "events around the launch (bench.py); counters = rocprofv3 --pmc over three training steps "
             "(tools/traffic_probe.py), per wave and per time step (T = %d)." % (workload, T), "",
             "| variant | kernel ms | step ms | VALU insts / step | VALU-active cycles / step | wave cycles / step | "
             "parked (s_waitcnt, barrier) | issue stalls | LDS insts / step | MFMA insts / step | MFMA busy cycles / step | "
             "MFMA and VALU together | v_exp etc. / step |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for v in VARIANTS:
        try:
            b = json.load(open(os.path.join(src, "bench_v%d.json" % v)))
            a = json.load(open(os.path.join(src, "pmc_a_v%d.json" % v)))
            m = json.load(open(os.path.join(src, "pmc_b_v%d.json" % v)))
        except (OSError, ValueError) as e:
            lines.append("| %s | (missing: %s) |" % (NAMES[v], e))
            continue
        w, w2 = max(a["SQ_WAVES"], 1.0), max(m["SQ_WAVES"], 1.0)
        wc = 4.0 * a["SQ_WAVE_CYCLES"] / w / T
        lines.append("| %s | %.3f | %.3f | %.0f | %.0f | %.0f | %.0f %% | %.0f %% | %.0f | %.1f | %.0f | %.0f | %.0f |" % (
            NAMES[v], b["config"]["native_ms_per_step"]["psvo_bsim_backward"], b["ms_per_step"],
            a["SQ_INSTS_VALU"] / w / T, 4.0 * a["SQ_ACTIVE_INST_VALU"] / w / T, wc,
            100.0 * a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 100.0 * a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
            a["SQ_INSTS_LDS"] / w / T, (m["SQ_INSTS_VALU_MFMA_F32"] + m.get("SQ_INSTS_VALU_MFMA_BF16", 0.0)) / w2 / T, m["SQ_VALU_MFMA_BUSY_CYCLES"] / w2 / T,
            m["SQ_VALU_MFMA_COEXEC_CYCLES"] / w2 / T, m["SQ_INSTS_VALU_TRANS_F32"] / w2 / T))
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3])
    else:
        report(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "C*")
