#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per (kernel, counter).

    python tools/pmc_summary.py gpurun_out/pmc_fit/*_counter_collection.csv [--match fit_columns]
Prints JSON: {kernel: {counter: {"sum": total over dispatches, "dispatches": n}}}.
"""
import csv
import json
import sys
from collections import defaultdict


def main() -> None:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    for path in args:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                if match and match not in k:
                    continue
                k = k.split("(")[0]
                c = acc[k][row["Counter_Name"]]
                c[0] += float(row["Counter_Value"])
                c[1].add(row["Dispatch_Id"])
    out = {k: {c: {"sum": v[0], "dispatches": len(v[1])} for c, v in cs.items()} for k, cs in acc.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
