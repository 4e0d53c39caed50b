"""Scratch loads / stores per basic block of one render kernel in build/nwe_kernel_mfma.s (make -C csrc asm)."""
import re, sys
want = sys.argv[1] if len(sys.argv) > 1 else "ILi256ELi8ELi4ELb1ELb0ELb1"
lines = open("nerf-workspaces-explorer_amd/csrc/build/nwe_kernel_mfma.s").read().split("\n")
name = lab = None
stats, order = {}, []
for l in lines:
    m = re.match(r"^(_ZN3nwe18render_mfma_kernel\S*):", l)
    if m:
        name = m.group(1); continue
    if name is None or want not in name: continue
    if "s_endpgm" in l: break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or lab is None:
        lab = m.group(1) if m else "entry"; order.append(lab); stats[lab] = dict(ld=0, st=0, ldB=0, stB=0, mfma=0, n=0, glds=0)
        if m: continue
    t = l.strip()
    if not t or t[0] in ";.": continue
    st = stats[lab]; st["n"] += 1
    w = {"dword ": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16}
    for k, v in w.items():
        if "scratch_load_" + k.strip() + " " in t + " ": pass
    mm = re.match(r"scratch_(load|store)_dword(x\d)?", t)
    if mm:
        b = 4 * int(mm.group(2)[1]) if mm.group(2) else 4
        if mm.group(1) == "load": st["ld"] += 1; st["ldB"] += b
        else: st["st"] += 1; st["stB"] += b
    if "v_mfma" in t: st["mfma"] += 1
    if "global_load_lds" in t: st["glds"] += 1
tot = dict(ld=0, st=0)
for lab in order:
    st = stats[lab]
    if st["ld"] or st["st"] or st["mfma"] > 20:
        print(lab, st)
