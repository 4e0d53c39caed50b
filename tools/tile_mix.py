"""Instruction mix per 48-MFMA tile in the hot basic blocks of one render kernel (build/nwe_kernel_mfma.s from `make asm`,
or build/one.s from `make one`): with one wave per SIMD every instruction beside an MFMA is an issue slot."""
import collections, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "nerf-workspaces-explorer_amd/csrc/build/one.s"
want = sys.argv[2] if len(sys.argv) > 2 else "ILi256ELi8ELi4ELb1ELb0ELb1ELb1"
lines = open(path).read().split("\n")
name = lab = None
blocks = collections.OrderedDict()
for l in lines:
    m = re.match(r"^(_ZN3nwe18render_mfma_kernel\S*):", l)
    if m:
        name = m.group(1); continue
    if name is None or want not in name: continue
    if "s_endpgm" in l: break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        lab = m.group(1); blocks[lab] = []; continue
    if lab is None: continue
    t = l.strip()
    if not t or t[0] in ";.": continue
    blocks[lab].append(t.split()[0])
tot = collections.Counter(); ntiles = 0
for lab, ins in blocks.items():
    n = sum(1 for i in ins if i.startswith("v_mfma"))
    if n < 200: continue
    c = collections.Counter()
    for i in ins:
        k = ("mfma" if i.startswith("v_mfma") else "accvgpr" if i.startswith("v_accvgpr") else "valu" if i.startswith("v_") else
             "lds" if i.startswith("ds_") else "dma" if i.startswith("global_load_lds") else i if i in ("s_waitcnt", "s_nop", "s_barrier") else
             "salu" if i.startswith("s_") else "other")
        c[k] += 1
    tiles = n / 48
    side = sum(v for k, v in c.items() if k != "mfma")
    print(lab, f"{n} mfma = {tiles:.1f} tiles; per tile:", {k: round(v / tiles, 1) for k, v in c.items() if k != "mfma"}, f"side/mfma {side / n:.2f}")
    vc = collections.Counter(i for i in ins if i.startswith("v_") and not i.startswith("v_mfma"))
    print("    valu:", {k: round(v / tiles, 1) for k, v in vc.most_common(10)})
    sc = collections.Counter(i for i in ins if i.startswith("s_") and i not in ("s_waitcnt", "s_nop", "s_barrier"))
    print("    salu:", {k: round(v / tiles, 1) for k, v in sc.most_common(8)})
