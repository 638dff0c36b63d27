#!/usr/bin/env python3
"""profiles/r0N_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) over the same command: the
counters of every kernel summed per STAGE of a batch and divided by the launches, for bench.py's roofline.stages[].traffic.
usage: python tools/pmc_traffic.py <fetch dir> <write dir> <out.json> <frames per batch launch> <frames per LfGroup launch> <commit> <command>"""
import collections, csv, glob, json, re, sys

STAGES = [("LfGroup streams", r"k_lf_rows|k_lf_lanes|k_lf_predict", "lf"), ("plan build + LfGroup tail", r"k_plan_|k_lf_dequant|k_llf_|k_clear_block_events", "batch"),
          ("entropy decode", r"k_hf_lanes|k_hf_entropy", "batch"), ("pixels", r"k_vardct_|k_k2_tiles", "batch")]


def totals(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    return tot, n


def main(fetch_dir, write_dir, out, batch_frames, lf_frames, commit, command):
    ft, fn = totals(fetch_dir, "FETCH_SIZE")
    wt, wn = totals(write_dir, "WRITE_SIZE")
    def launches(n, pat):
        return max([c for k, c in n.items() if re.search(pat, k)] or [0])
    res = {"stream": "forward", "frame": "7680x4320, tools/jxlsynth forward=1, seeds 3 + 1000 i", "commit": commit, "command": command, "fetch_correction": 2.0, "stages": {}, "kernels": {}}
    for name, pat, kind in STAGES:
        lf, lw = launches(fn, r"k_lf_rows|k_lf_lanes") if kind == "lf" else launches(fn, r"k_hf_lanes"), launches(wn, r"k_lf_rows|k_lf_lanes") if kind == "lf" else launches(wn, r"k_hf_lanes")
        if not lf or not lw:
            continue
        res["stages"][name] = {"fetch_kb": round(sum(v for k, v in ft.items() if re.search(pat, k)) / lf, 1), "write_kb": round(sum(v for k, v in wt.items() if re.search(pat, k)) / lw, 1),
                               "frames_per_launch": float(lf_frames if kind == "lf" else batch_frames), "launches_counted": [lf, lw]}
    for k in sorted(set(ft) | set(wt)):
        res["kernels"][k[:90]] = {"fetch_kb_per_dispatch": round(ft.get(k, 0) / max(fn.get(k, 1), 1), 1), "write_kb_per_dispatch": round(wt.get(k, 0) / max(wn.get(k, 1), 1), 1), "dispatches": [fn.get(k, 0), wn.get(k, 0)]}
    res["source"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `%s` on commit %s (tools/r06_final.sh; round 5: tools/r05_final.sh); KB as reported, per launch: every kernel of a stage summed, divided by the stage's "
                     "launches; FETCH_SIZE is doubled by bench.py per the gfx950 note in MI355X_MICROARCH.md (128-byte requests tallied at 64) -- calibrated there for wide coalesced reads, an upper bound for narrow ones; "
                     "under --pmc the kernels run one at a time, the LfGroup launches carry what was waiting then") % (command, commit)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["stages"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), float(sys.argv[5]), sys.argv[6], sys.argv[7])
