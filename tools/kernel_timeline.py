#!/usr/bin/env python3
"""a rocprofv3 --kernel-trace run (rocpd database) as a timeline: every dispatch of at least MIN_MS (default 1) with its start, its
duration, its queue and a short name, in start order -- who runs beside whom in the pipeline's steady state -- and, per 50 ms
window, how many of j40hip's stages had a kernel running.
usage: python tools/kernel_timeline.py <dir with *.db> out.txt [min ms] [skip ms from the first dispatch] [at most ms]"""
import glob, re, sqlite3, sys


def short(name):
    m = re.match(r"_ZN6j40hip(\d+)", name)   # (rocpd keeps mangled names: namespace j40hip, then the length-prefixed kernel name)
    if m:
        at = m.end(); n = int(m.group(1)); rest = name[at + n:]
        t = re.match(r"I((?:L[ib]\d+E)+)E", rest)   # template arguments that are integer / bool literals
        return (name[at:at + n] + ("<" + ",".join(re.findall(r"L[ib](\d+)E", t.group(1))) + ">" if t else ""))[:44]
    n = name.replace("j40hip::", "").replace("void ", "")
    n = n.split("(")[0]
    return n[:44]


def stage_of(n):
    if "k_lf_rows" in n or "k_lf_lanes" in n: return "L"
    if "k_hf_lanes" in n: return "E"
    if "k_vardct" in n: return "P"
    if "k_plan" in n or "k_lf_dequant" in n or "k_llf" in n or "k_clear_block" in n: return "B"
    return None


def main(path, out, min_ms=1.0, skip_ms=0.0, span_ms=1e9):
    dbs = glob.glob(path + "/**/*.db", recursive=True) + glob.glob(path + "/*.db")
    db = sqlite3.connect(dbs[0])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % disp)]
    qcol = "queue_id" if "queue_id" in cols else None
    scol = "stream_id" if "stream_id" in cols else None
    sel = "d.start, d.end, s.kernel_name" + (", d.%s" % qcol if qcol else ", 0") + (", d.%s" % scol if scol else ", 0")
    rows = sorted(db.execute("select %s from %s d join %s s on d.kernel_id=s.id" % (sel, disp, sym)))
    if not rows:
        open(out, "w").write("no dispatches\n"); return
    t0 = rows[0][0]
    lines = ["# columns of %s: %s" % (disp, ", ".join(cols)), "# start_ms  dur_ms  queue stream  kernel   (dispatches of at least %.1f ms, from %.0f ms on)" % (min_ms, skip_ms)]
    queues = {}
    for (s, e, n, q, st) in rows:
        a = (s - t0) / 1e6
        if a < skip_ms or a > skip_ms + span_ms or (e - s) / 1e6 < min_ms: continue
        qi = queues.setdefault(q, len(queues))
        lines.append("%9.2f %7.2f  q%-2d s%-3s %s" % (a, (e - s) / 1e6, qi, st, short(n)))
    # per window: the share of the window during which each stage had at least one kernel running
    win = 50.0
    last = (rows[-1][1] - t0) / 1e6
    lines.append("# per %d ms window: share of the window with a kernel of the stage running (L LfGroup streams, B plan build + tail, E entropy, P pixels), and with none at all" % win)
    w = skip_ms
    while w < min(last, skip_ms + span_ms):
        cover = {}
        iv_all = []
        for (s, e, n, q, st) in rows:
            a, b = (s - t0) / 1e6, (e - t0) / 1e6
            if b <= w or a >= w + win: continue
            k = stage_of(n)
            iv = (max(a, w), min(b, w + win))
            iv_all.append(iv)
            if k: cover.setdefault(k, []).append(iv)

        def union(ivs):
            tot = 0.0; cur_a = cur_b = None
            for a, b in sorted(ivs):
                if cur_b is None or a > cur_b:
                    if cur_b is not None: tot += cur_b - cur_a
                    cur_a, cur_b = a, b
                else: cur_b = max(cur_b, b)
            if cur_b is not None: tot += cur_b - cur_a
            return tot
        lines.append("%8.0f  " % w + "  ".join("%s %3.0f%%" % (k, 100 * union(cover.get(k, [])) / win) for k in "LBEP") + "  idle %3.0f%%" % (100 * (1 - union(iv_all) / win)))
        w += win
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:6]))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2], float(a[3]) if len(a) > 3 else 1.0, float(a[4]) if len(a) > 4 else 0.0, float(a[5]) if len(a) > 5 else 1e9)
