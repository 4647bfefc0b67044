"""bench.py's measured fractions come from the committed counter passes: the kernel names it looks up must be in them."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_kernels_named_by_bench_have_committed_counters():
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    counters = json.load(open(bench.PMC_FILE))["C3"]
    named = set(re.findall(r'"(k_[a-z_]+<[^"]*>)"', src))
    assert {"k_shade_candidates<false, true, false>", "k_search_primary<4>"} <= named
    for k in named:
        assert k in counters, f"{k} is not in {bench.PMC_FILE}: {sorted(counters)}"
        frac = bench.pmc_fractions(counters[k])
        assert frac and 0 < frac["valu_issue_frac"] < 1
        for key, v in frac.items():
            if "frac" in key and isinstance(v, float):
                assert 0 <= v <= 1, (k, key, v)


def test_bench_line_on_file_is_a_fraction_free_of_values_above_one():
    """The committed bench line: every fraction in it is at most 1, the headline is traced rays."""
    line = open(os.path.join(ROOT, "profiles", "r02_bench_c3.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    assert d["unit"] == "Mrays/s" and d["roofline"]["frac"] <= 1 and d["vs_baseline"] is None

    def walk(x, path=""):
        if isinstance(x, dict):
            for k, v in x.items():
                walk(v, path + "/" + k)
        elif isinstance(x, float) and "frac" in path.rsplit("/", 1)[-1]:
            assert 0 <= x <= 1, (path, x)
    walk(d)
