#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace CSV into a timeline of the LAST factorisation in it:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/potrf_once.py 8192 3
   python3 tools/timeline.py gpurun_out/trace > profiles/rNN_timeline_n8192.txt
columns: start_us end_us duration_us queue kernel [grid workgroups]"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.search(r"k_gemm_nt_sub<(\w+), (true|false), (\d)>", name)
    if m:
        return "k_gemm_nt_sub<%s,%s-tile>" % ("lower" if m.group(2) == "true" else "rect", 32 * int(m.group(3)))
    name = re.sub(r"cimrgp::\(anonymous namespace\)::", "", name)
    name = re.sub(r"<.*", "", name)
    return name.split("(")[0][:48]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + root)
    rows = []
    with open(sorted(files)[-1]) as fh:
        for r in csv.DictReader(fh):
            rows.append(dict(name=r["Kernel_Name"], q=r.get("Queue_Id", "?"), s=int(r["Start_Timestamp"]),
                             e=int(r["End_Timestamp"]),
                             wg=(int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1)))))
    rows.sort(key=lambda r: r["s"])
    # the last factorisation = from the last Gram build on
    start = 0
    for i, r in enumerate(rows):
        if "k_rbf_gram" in r["name"]:
            start = i
    if len(sys.argv) > 2 and sys.argv[2] == "boundary":
        # a window around the LAST Gram build (pipelined steps: the previous step's tail and solve stage beside the
        # next step's front end and first panels): 400 us before it to 900 us after it
        # (argv[3] = K: around the K-th symmetric Gram build instead -- the timed steps of bench.py come before its
        # stage and potrf-alone sections, which build Gram matrices too)
        if len(sys.argv) > 3:
            grams = [r for r in rows if "k_rbf_gram" in r["name"]]
            big = max(r["wg"] for r in grams)
            sym = [r for r in grams if r["wg"] == big]
            tg = sym[int(sys.argv[3]) - 1]["s"]
        else:
            tg = rows[start]["s"]
        before_us = int(sys.argv[4]) if len(sys.argv) > 4 else 400     # (argv[4], argv[5]: the window in us before / after it)
        after_us = int(sys.argv[5]) if len(sys.argv) > 5 else 900
        rows = [r for r in rows if tg - before_us * 1000 <= r["s"] <= tg + after_us * 1000]
        t0 = tg
        print("# window around the last Gram build (time 0): the previous step's tail and solve stage | the next step's front end and first panels")
    else:
        rows = rows[start:]
        t0 = rows[0]["s"]
    queues = {}
    print("# columns: start_us end_us duration_us queue kernel workgroups")
    print("# total span: %.1f us" % ((max(r["e"] for r in rows) - t0) / 1e3))
    # one stand-alone diagonal kernel per panel (its first 64-column block; the other three are factored
    # inside the link kernels) unless the chain runs as round 1's 4 x (k_diag64, k_trsm64)
    diag = [r["s"] for r in rows if "k_diag64" in r["name"]]
    links = any("k_link" in r["name"] for r in rows)
    per_panel = diag if links else diag[::4]
    print("# period between consecutive panels (first diagonal kernel to first diagonal kernel, us): " +
          " ".join("%.0f" % ((b - a) / 1e3) for a, b in zip(per_panel[:-1], per_panel[1:])))
    for r in rows:
        q = queues.setdefault(r["q"], "q%d" % (len(queues) + 1))
        print("%9.1f %9.1f %7.1f  %s %s %d" % ((r["s"] - t0) / 1e3, (r["e"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, q,
                                             short(r["name"]), r["wg"]))


if __name__ == "__main__":
    main()
