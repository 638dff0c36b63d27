"""the measurement helpers that run without a GPU: the kernel-trace timeline (tools/kernel_timeline.py) over a rocpd database made
here, bench.py's reading of the container's CPU accounting"""
import importlib.util
import os
import sqlite3
import sys

from streams import ROOT


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, [path]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_kernel_timeline_names_and_windows(tmp_path):
    kt = load(os.path.join(ROOT, "tools", "kernel_timeline.py"), "kernel_timeline")
    assert kt.short("_ZN6j40hip12k_vardct_dctILi3ELi3ELi16ELb1EEEvNS_7DevPlanE") == "k_vardct_dct<3,3,16,1>"
    assert kt.short("_ZN6j40hip10k_hf_lanesEPKNS_7DevPlanE") == "k_hf_lanes" and kt.stage_of("_ZN6j40hip10k_hf_lanesEPKNS_7DevPlanE") == "E"
    assert kt.stage_of("_ZN6j40hip9k_lf_rowsILb0EEEvPKNS_12DevLfLaneSetE") == "L" and kt.stage_of("_ZN6j40hip12k_plan_placeEPK") == "B" and kt.stage_of("__amd_rocclr_copyBuffer") is None
    db = sqlite3.connect(str(tmp_path / "x.db"))
    db.execute("create table rocpd_kernel_dispatch_t (id integer, kernel_id integer, start integer, end integer, queue_id integer, stream_id integer)")
    db.execute("create table rocpd_info_kernel_symbol_t (id integer, kernel_name text)")
    db.execute("insert into rocpd_info_kernel_symbol_t values (0, '_ZN6j40hip10k_hf_lanesEPKNS_7DevPlanE')")
    db.execute("insert into rocpd_info_kernel_symbol_t values (1, '_ZN6j40hip20k_vardct_special_afvILi16ELb1EEEvNS_7DevPlanE')")
    t0 = 10 ** 12
    db.execute("insert into rocpd_kernel_dispatch_t values (0, 0, ?, ?, 7, 1)", (t0, t0 + 25_000_000))                  # 0 .. 25 ms
    db.execute("insert into rocpd_kernel_dispatch_t values (1, 1, ?, ?, 9, 2)", (t0 + 20_000_000, t0 + 60_000_000))     # 20 .. 60 ms
    db.execute("insert into rocpd_kernel_dispatch_t values (2, 1, ?, ?, 9, 2)", (t0 + 60_100_000, t0 + 60_400_000))     # 0.3 ms: below the threshold
    db.commit(); db.close()
    out = tmp_path / "timeline.txt"
    kt.main(str(tmp_path), str(out), 1.0, 0.0)
    lines = out.read_text().splitlines()
    rows = [l for l in lines if not l.startswith("#") and " q" in l]
    assert len(rows) == 2 and "k_hf_lanes" in rows[0] and "k_vardct_special_afv<16,1>" in rows[1] and "q0" in rows[0] and "q1" in rows[1]
    first, second = [l for l in lines if l.strip().startswith(("0 ", "50 "))][:2]
    assert "E  50%" in first and "P  60%" in first and "idle   0%" in first      # 0 .. 50 ms: E 0 .. 25, P 20 .. 50, nothing uncovered
    assert "E   0%" in second and "P  21%" in second and "idle  79%" in second   # 50 .. 100 ms: P 10 + 0.3 ms


def test_bench_reads_the_containers_cpu_accounting():
    bench = load(os.path.join(ROOT, "bench.py"), "bench_for_tools_test")
    st = bench.cgroup_cpu_stat()
    assert st is None or (st["usage_usec"] > 0 and "throttled_usec" in st)
    assert bench.cpu_quota() >= 1
