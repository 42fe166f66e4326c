#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of tools/profile_round.sh into the per-round summaries bench.py and DESIGN.md cite:

    gpurun_out/prof_<tag>/<tag>_<wl>_kernel_stats.csv     (copy of the --stats table)
    gpurun_out/prof_<tag>/<tag>_<wl>_pmc_traffic.json     HBM bytes of the dominant score kernel per launch
    gpurun_out/prof_<tag>/<tag>_<wl>_score_counters.json  SQ / LDS / TCP / TCC counters of that kernel per launch
    gpurun_out/prof_<tag>/<tag>_<wl>_fit_pmc.json         the same for the fit kernels
Copy them into profiles/ to have them judged.   usage: pmc_round_summary.py <dir> <tag> <workload>
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def load(d, wl):
    """Per kernel and counter: (mean per launch, launches) over the launches of the kernel's LARGEST grid -- a bench run also
    launches the score kernels for small row sets (flagged rows, warm-ups of other modes), which are not the pass priced here."""
    rows = []
    for path in glob.glob(os.path.join(d, f"pmc*_{wl}_counter_collection.csv")):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                row["_k"] = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                rows.append(row)
    top_grid = defaultdict(int)
    for row in rows:
        top_grid[row["_k"]] = max(top_grid[row["_k"]], int(row.get("Grid_Size") or 0))
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    for row in rows:
        k = row["_k"]
        if int(row.get("Grid_Size") or 0) != top_grid[k]:
            continue
        c = acc[k][row["Counter_Name"]]
        c[0] += float(row["Counter_Value"])
        c[1].add(row["Dispatch_Id"])
    return {k: {c: (v[0] / max(len(v[1]), 1), len(v[1])) for c, v in cs.items()} for k, cs in acc.items()}


def build_stamp():
    """Which library build the counters describe (rtrec_amd.build.fingerprint: sha256 of librtrec_amd.so + the git commit it
    was built at); bench.py attaches a summary to its line only when this matches the build it is timing."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from rtrec_amd import build
        return build.fingerprint()
    except Exception as exc:
        return {"lib_sha256": None, "git_head": None, "error": repr(exc)}


def main():
    d, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    per = load(d, wl)
    stamp = build_stamp()
    stats = glob.glob(os.path.join(d, f"stats_{wl}_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(d, f"{tag}_{wl}_kernel_stats.csv"))
    # The kernel summarised is the one the bench line of the same run names as dominant (roofline.kernel), else the score
    # kernel with the most wave cycles over ALL candidates -- never "any segment kernel first": the C3 run also launches
    # score_seg_kernel for a few request-sized calls, and round 3's summaries described those.
    score = [k for k in per if any(n in k for n in ("score_seg_kernel", "score_frows_kernel", "score_sparse_kernel<float, false>"))]
    named = None
    try:
        named = json.load(open(os.path.join(d, f"bench_{wl}.json")))["roofline"]["kernel"].split("<")[0].split(" ")[0]
    except Exception:
        pass
    if named and [k for k in score if named in k]:
        score = [k for k in score if named in k]
    if score:
        k = max(score, key=lambda n: per[n].get("SQ_WAVE_CYCLES", (0, 0))[0] * per[n].get("SQ_WAVE_CYCLES", (0, 0))[1] or
                per[n].get("FETCH_SIZE", (0, 0))[0])
        c = {n: v[0] for n, v in per[k].items()}
        heavy = [h for h in per if "score_seg_heavy_kernel" in h] if "score_seg_kernel" in k else []
        if heavy:       # the segment path is two launches per pass (long users first): byte counters are summed, the rest kept apart
            hc = {n: v[0] for n, v in per[heavy[0]].items()}
            for n in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
                c[n] = c.get(n, 0.0) + hc.get(n, 0.0)
            c.update({"heavy_" + n: v for n, v in hc.items() if n.startswith("SQ_")})
        launches = max(v[1] for v in per[k].values())
        fetch_kb, write_kb = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
        hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        json.dump({"build": stamp, "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes, "
                             f"tools/profile_round.sh {tag} {wl}) -- python3 bench.py --steps 3 --no-cpu-baseline",
                   "kernel": k.replace("void rtrec::", "").replace("void ", "").replace(" ", ""), "launches_averaged": launches,
                   "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
                   "hbm_bytes_per_launch_corrected": int(fetch_kb * 1024 * 2 + write_kb * 1024),
                   "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
                   "note": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> doubled per MI355X_MICROARCH.md; "
                           "Infinity-Cache hits are counted, so this is an upper bound of DRAM traffic"},
                  open(os.path.join(d, f"{tag}_{wl}_pmc_traffic.json"), "w"), indent=1)
        wave_cyc = c.get("SQ_WAVE_CYCLES", 0.0)
        json.dump({"build": stamp, "kernel": k.replace("void rtrec::", "").replace("void ", "").replace(" ", ""), "launches_averaged": launches,
                   "per_launch": c,
                   "derived": {"valu_busy_frac_of_wave_cycles": c.get("SQ_ACTIVE_INST_VALU", 0.0) / wave_cyc if wave_cyc else None,
                               "wait_any_frac": c.get("SQ_WAIT_ANY", 0.0) / wave_cyc if wave_cyc else None,
                               "lds_bank_conflict_frac_of_lds_cycles": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"])
                               if c.get("SQ_LDS_IDX_ACTIVE") else None,
                               "l1_miss_rate": (c.get("TCP_TCC_READ_REQ_sum", 0.0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"])
                               if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None},
                   "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"},
                  open(os.path.join(d, f"{tag}_{wl}_score_counters.json"), "w"), indent=1)
    fit = {k: v for k, v in per.items() if "fit_columns" in k}
    if fit:
        out = {}
        for k, cs in fit.items():
            c = {n: v[0] for n, v in cs.items()}
            fetch_kb, write_kb = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
            hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
            out[k.replace("void rtrec::", "").replace("void ", "").replace(" ", "")] = {
                "launches_averaged": max(v[1] for v in cs.values()), "per_launch": c,
                "hbm_bytes_per_launch_corrected": int(fetch_kb * 1024 * 2 + write_kb * 1024),
                "l2_hit_rate": hit / (hit + miss) if hit + miss else None}
        json.dump(dict(out, build=stamp, note="FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); the fit's "
                                 "residual gathers are 4-byte accesses, for which the counter is uncalibrated (+-2x on the read side)"),
                  open(os.path.join(d, f"{tag}_{wl}_fit_pmc.json"), "w"), indent=1)
    print("summaries:", sorted(f for f in os.listdir(d) if f.startswith(tag)))


if __name__ == "__main__":
    main()
