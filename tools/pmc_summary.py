"""Summarise rocprofv3 --pmc CSV output for the render kernel (last dispatch): per-wave cycle shares."""
import csv, collections, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*_counter_collection.csv") + glob.glob(f"{d}/*/*_counter_collection.csv"):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "render_mfma" not in r["Kernel_Name"]:
                continue
            key = (int(r["Dispatch_Id"]), r["Counter_Name"])
            agg[key] = agg.get(key, 0) + float(r["Counter_Value"])
        if not agg:
            continue
        last = max(k[0] for k in agg)
        c = {k[1]: v for k, v in agg.items() if k[0] == last}
        print(d, {k: f"{v:.4g}" for k, v in c.items()})
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            for k, v in c.items():
                if k not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"):
                    print(f"   {k}: {v / wc:.2%} of wave cycles")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                print(f"   MFMA busy: {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc):.1%} of wave cycles (1 wave per SIMD)")
