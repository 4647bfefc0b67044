"""bench.py's measured fractions come from the committed counter passes: the kernel names it looks up must be in them,
the passes must say which commit they were taken at, and the launch they profiled must still be the launch the
committed bench line timed (within 10 %)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BENCH_LINE = os.path.join(ROOT, "profiles", "r03_bench_c3.json")


def bench_line(path=BENCH_LINE):
    return json.loads(open(path).read().strip().splitlines()[-1])


def test_kernels_named_by_bench_have_committed_counters():
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    counters = json.load(open(bench.PMC_FILE))["C3"]
    named = set(re.findall(r'"(k_[a-z_]+<[^"]*>)"', src))
    assert {"k_search_tree_fast<4>", "k_shade_candidates<false, true, false>", "k_search_primary<4>"} <= named
    named.add("k_shade_candidates_fast")  # (no template arguments: the pattern above does not see it)
    for k in named:
        assert k in counters, f"{k} is not in {bench.PMC_FILE}: {sorted(counters)}"
        assert re.fullmatch(r"[0-9a-f]{7,40}", counters[k]["commit"]), counters[k].get("commit")
        frac = bench.pmc_fractions(counters[k])
        assert frac and 0 < frac["valu_issue_frac"] < 1
        assert frac["replayed_from"] == os.path.relpath(bench.PMC_FILE, ROOT) and frac["profiled_commit"] == counters[k]["commit"]
        for key, v in frac.items():
            if "frac" in key and isinstance(v, float):
                assert 0 <= v <= 1, (k, key, v)


def test_replayed_counters_call_themselves_stale_when_the_launch_time_moved():
    import bench
    k = json.load(open(bench.PMC_FILE))["C3"]["k_search_tree_fast<4>"]
    ms = k["avg_launch_ms"]
    assert bench.pmc_fractions(k, ms * 1.05)["stale"] is False
    assert bench.pmc_fractions(k, ms * 1.12)["stale"] is True
    assert bench.pmc_fractions(k, ms * 0.85)["stale"] is True
    assert "stale" not in bench.pmc_fractions(k)


def test_committed_bench_line_was_timed_on_the_launches_that_were_profiled():
    """Every `executed` object of the committed line carries this run's launch time against the profiled one."""
    d = bench_line()
    seen = []

    def walk(x, path=""):
        if isinstance(x, dict):
            if "avg_launch_ms_profiled" in x:
                assert "launch_ms_drift_frac" in x, path
                assert x["launch_ms_drift_frac"] <= 0.10 and x["stale"] is False, (path, x["launch_ms_drift_frac"])
                assert x["replayed_from"].startswith("profiles/r03_") and x["profiled_commit"], path
                seen.append(path)
            for k, v in x.items():
                walk(v, path + "/" + k)
    walk(d)
    assert {"/roofline/executed", "/roofline/second_kernel/executed", "/list_scan/executed"} <= set(seen), seen


def test_bench_line_on_file_is_a_fraction_free_of_values_above_one():
    """The committed bench line: every fraction in it is at most 1, the headline is traced rays, and the legs that the
    driver's single run must also time (form T, an extension-flag frame) are keys of it."""
    d = bench_line()
    assert d["unit"] == "Mrays/s" and d["roofline"]["frac"] <= 1 and d["vs_baseline"] is None
    both = {d["roofline"]["kernel"], d["roofline"]["second_kernel"]["kernel"]}
    assert both == {"k_search_tree_fast<4>", "k_shade_candidates_fast"} and d["roofline"]["bound"] == "valu_fp32"
    assert d["roofline"]["avg_launch_ms"] >= d["roofline"]["second_kernel"]["avg_launch_ms"]  # the roofline is the longer kernel's
    for key in ("form_T", "shadows_enter_bounded", "static_camera", "list_scan", "cpu_baseline", "fp32_first", "work_buffer_bytes_rank0"):
        assert key in d, key
    assert d["form_T"]["ms_per_step"] > 0 and d["shadows_enter_bounded"]["ms_per_step"] > 0
    share = d["fp32_first"]
    assert 0 <= share["share_of_hit_pixels_shaded_again"] < 0.05 and share["share_of_hits_decided_in_fp32"] > 0.9, share

    def walk(x, path=""):
        if isinstance(x, dict):
            for k, v in x.items():
                walk(v, path + "/" + k)
        elif isinstance(x, float) and "frac" in path.rsplit("/", 1)[-1]:
            assert 0 <= x <= 1, (path, x)
    walk(d)


def test_committed_c5_line_checks_every_frame_digest():
    d = bench_line(os.path.join(ROOT, "profiles", "r03_bench_c5.json"))
    assert d["config"]["workload"].startswith("C5") and d["config"]["partition"].startswith("whole frames")
    equal, _, checked = d["frames_equal_to_oracle_digest"].split(" ")[:3]
    assert int(equal) == int(checked) >= 3
    assert len(d["frames"]) == d["steps"] and all(len(f["sha256"]) == 64 for f in d["frames"])
