"""profiles/r02_shade_before_after.txt: the FP64 shading launch of the C3 frame in the two committed sets of counter passes
(profiles/r01_*: hits shaded from the shade queue with their secondary rays traced; profiles/r02_*: shaded from the
candidate lists)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(path, kernel):
    for line in open(path):
        if line.startswith(kernel + " "):
            f = line[len(kernel):].split()
            return int(f[0]), int(f[3]), int(f[5])  # VGPRs, scratch bytes, waves per SIMD
    raise KeyError(kernel)


def main():
    a = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_shade_kernel.json")))
    b = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_kernels.json")))["C3"]
    r1 = os.path.join(ROOT, "profiles", "r01_kernel_resources.txt")
    r2 = os.path.join(ROOT, "profiles", "r02_kernel_resources.txt")

    def row(name, res, ms, k, waits=True):
        busy = k["SQ_ACTIVE_INST_VALU"] * 4 / (k["GRBM_GUI_ACTIVE"] / 8 * 1024)
        w = f"{k['SQ_WAIT_ANY'] / k['SQ_WAVE_CYCLES']:.2f}" if waits else "n/a"
        return (f"{name:<62} {res[0]:>5} {res[1]:>7} {res[2]:>5} {ms:>8.3f} {k['SQ_INSTS_VALU'] / 1e6:>10.1f} {k['SQ_INSTS_SALU'] / 1e6:>10.1f} "
                f"{busy:>9.3f} {w:>6}")

    full, inert = "k_shade_candidates<false, false, false>", "k_shade_candidates<false, true, false>"
    lines = [
        "Shading launch of the C3 frame (4K, 16 spp, 3 lights, form B), the r01 profiles against the r02 profiles, from the rocprofv3 --pmc passes",
        "(profiles/r01_pmc_shade_kernel.json, profiles/r02_pmc_kernels.json) and the compiler's resource reports (r01/r02_kernel_resources.txt).",
        "VALU/SALU = wave-instructions per launch in millions; VALU busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs);",
        "wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of resident wave-cycles spent waiting; the r01 passes did not collect the two counters).",
        "The r02 launches also decide the rays' single candidates with Triangle.intersections (in r01 a kernel of its",
        "own, 0.34 ms, did that for every candidate): about 0.15 ms and 35 M VALU instructions of the figures below.",
        "",
        f"{'kernel':<62} {'VGPRs':>5} {'scratch':>7} {'waves':>5} {'ms':>8} {'VALU M':>10} {'SALU M':>10} {'VALU busy':>9} {'wait':>6}",
        row("r01: k_shade_candidates<false> (all hits, rays traced)", resources(r1, "k_shade_candidates<false>"), 1.54, a, waits=False),
        row("r02, full ray statistics: " + full, resources(r2, full), b[full]["avg_launch_ms"], b[full]),
        row("r02, pixels only:         " + inert, resources(r2, inert), b[inert]["avg_launch_ms"], b[inert]),
    ]
    out = os.path.join(ROOT, "profiles", "r02_shade_before_after.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
