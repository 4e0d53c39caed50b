"""Turn the output of tools/r03_profile.sh (gpurun_out/r03_*) into the committed profiles/r03_* files:
kernel stats CSV, counter summary, and the traffic JSON bench.py reads (tied to the kernel sources by hash)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def counters(d):
    """{counter: mean per FRAME} of one --pmc pass: a frame is one dispatch of the packets instantiation of render_mfma_kernel
    plus, under the hybrid launch plan, one of the sample-split instantiation right behind it (template argument 5); the
    counters of both are summed."""
    agg = collections.defaultdict(float)
    frames, launches = set(), set()
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "render_mfma" not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            launches.add(r["Dispatch_Id"])
            args = r["Kernel_Name"].split("<", 1)[-1].split(",")
            if len(args) > 4 and args[4].strip() == "false":
                frames.add(r["Dispatch_Id"])
    n = max(len(frames) or len(launches), 1)
    return {k: v / n for k, v in agg.items()}, n


def main():
    import bench
    lines = []
    # 1. kernel trace
    stats = glob.glob(f"{OUT}/r03_prof/**/*kernel_stats.csv", recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(PROF, "r03_kernel_stats.csv"))
        per_frame = 0.0
        for r in csv.DictReader(open(stats[0])):
            if "render_mfma" in r["Name"]:
                lines.append(f"kernel trace: {r['Name'][:70]}: calls {r['Calls']}, average {float(r['AverageNs']) / 1e6:.2f} ms, "
                             f"{r['Percentage']} % of GPU time")
                per_frame += float(r["AverageNs"]) / 1e6
        lines.append(f"kernel trace: a frame = one launch of each of the render_mfma instantiations above (packets for the full rounds of "
                     f"workgroups + sample split for the ragged last round): {per_frame:.2f} ms per frame")
    for name in ("r03_bench_under_rocprof.json", "r03_bench.json"):
        if os.path.exists(os.path.join(OUT, name)):
            shutil.copy(os.path.join(OUT, name), os.path.join(PROF, name))
            d = json.load(open(os.path.join(OUT, name)))
            lines.append(f"{name}: kernel_ms (HIP events) {d['roofline']['kernel_ms']:.2f}, ms_per_step {d['ms_per_step']:.2f}, "
                         f"value {d['value']:.4g} {d['unit']}, frac {d['roofline']['frac']:.4f}, executed {d['roofline']['executed_mfma_tflops']:.0f} TFLOP/s")
    # 2. counters
    allc = {}
    for d in sorted(glob.glob(f"{OUT}/r03_pmc_*")):
        if os.path.isdir(d):
            c, n = counters(d)
            allc.update(c)
            lines.append(f"{os.path.basename(d)} ({n} frames): " + ", ".join(f"{k} {v:.4g}" for k, v in c.items()))
    if "SQ_WAVE_CYCLES" in allc:
        wc = allc["SQ_WAVE_CYCLES"]
        lines.append("per wave cycle (SQ_* cycle counters are in quad-cycles; one wave per SIMD):")
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if k in allc:
                lines.append(f"   {k}: {allc[k] / wc:.2%}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in allc:
            lines.append(f"   MFMA busy: {allc['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc):.1%} of wave cycles (SQ_VALU_MFMA_BUSY_CYCLES counts cycles)")
        if "SQ_INSTS_MFMA" in allc:
            lines.append(f"   one MFMA per {4 * wc / allc['SQ_INSTS_MFMA']:.1f} wave cycles (32 = the matrix pipe's rate); "
                         f"SQ_INSTS_MFMA per frame {allc['SQ_INSTS_MFMA']:.4g} (expected 640000/32 * (64+192) * 3120 = {640000 / 32 * 256 * 3120:.4g})")
    if "SQC_ICACHE_REQ" in allc:
        req, hit, miss = allc["SQC_ICACHE_REQ"], allc.get("SQC_ICACHE_HITS", 0.0), allc.get("SQC_ICACHE_MISSES", 0.0)
        lines.append(f"instruction fetch per frame: SQC_ICACHE_REQ {req:.4g}, hits {hit:.4g}, misses {miss:.4g} (+ duplicate {allc.get('SQC_ICACHE_MISSES_DUPLICATE', 0.0):.4g}) "
                     f"-> miss rate {miss / max(req, 1):.1e}; SQ_IFETCH {allc.get('SQ_IFETCH', 0.0):.4g}, mean fetches in flight per wave "
                     f"{allc.get('SQ_IFETCH_LEVEL', 0.0) / max(allc.get('SQ_WAVE_CYCLES', 1.0), 1):.3f}; i-cache busy {allc.get('SQC_ICACHE_BUSY_CYCLES', 0.0):.4g} cycles")
    if "GRBM_GUI_ACTIVE" in allc:
        kms = None
        try:
            kms = json.load(open(os.path.join(OUT, "r03_pmc_9.json")))["roofline"]["kernel_ms"]
        except Exception:   # noqa: BLE001
            pass
        lines.append(f"GRBM_GUI_ACTIVE per frame {allc['GRBM_GUI_ACTIVE']:.4g} (sum over 8 XCDs): effective clock = that / 8 / kernel time"
                     + (f" = {allc['GRBM_GUI_ACTIVE'] / 8 / (kms * 1e-3) / 1e9:.3f} GHz at {kms:.1f} ms in that pass" if kms else "")
                     + " (MI355X_MICROARCH.md, DVFS give-back; in-kernel s_memtime / s_memrealtime of the stamped build: profiles/r03_stamps_800.txt)")
    if "SQ_VALU_MFMA_COEXEC_CYCLES" in allc and "SQ_VALU_MFMA_BUSY_CYCLES" in allc:
        lines.append(f"   vector and matrix instructions executing together: {allc['SQ_VALU_MFMA_COEXEC_CYCLES'] / allc['SQ_VALU_MFMA_BUSY_CYCLES']:.1%} of the MFMA-busy cycles")
    if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
        # MI355X_MICROARCH.md, HBM section: both in KB; gfx950 tallies a 128-B request of a 16-B-per-lane stream at 64 B -> x2
        hbm = (2 * allc["FETCH_SIZE"] + allc["WRITE_SIZE"]) * 1024
        lines.append(f"traffic per frame: FETCH_SIZE {allc['FETCH_SIZE']:.4g} KB (x2 on gfx950), WRITE_SIZE {allc['WRITE_SIZE']:.4g} KB -> {hbm:.4g} B "
                     f"behind L2; algorithmic 1.76e7 B")
        if "TCC_HIT_sum" in allc:
            lines.append(f"   L2: hits {allc['TCC_HIT_sum']:.4g}, misses {allc['TCC_MISS_sum']:.4g} ({allc['TCC_HIT_sum'] / (allc['TCC_HIT_sum'] + allc['TCC_MISS_sum']):.2%} hit rate; "
                         f"misses x 128 B = {allc['TCC_MISS_sum'] * 128:.4g} B)")
        json.dump({"kernel_source_sha": bench.kernel_source_sha(), "hbm_bytes_per_launch": hbm,
                   "fetch_size_kb": allc["FETCH_SIZE"], "write_size_kb": allc["WRITE_SIZE"],
                   "note": "HBM-side bytes per C3 launch = (2 x FETCH_SIZE + WRITE_SIZE) KB from separate rocprofv3 --pmc passes over "
                           "bench.py (tools/r03_profile.sh); L2 misses of the L2->LDS weight stream, served by the Infinity Cache"},
                  open(os.path.join(PROF, "r03_pmc_traffic.json"), "w"), indent=1)
    open(os.path.join(PROF, "r03_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
