"""Where does a render kernel touch scratch?  Lists, per kernel of build/nwe_kernel_mfma.s (make -C csrc asm), the scratch
instructions with the number of MFMAs in front of them in program order, so spills inside the tile loop (between MFMAs) are
told apart from spills in the per-sample / per-pass code around it."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "nerf-workspaces-explorer_amd/csrc/build/nwe_kernel_mfma.s"
want = sys.argv[2] if len(sys.argv) > 2 else "ILi256ELi8ELi4ELb1ELb0ELb1"
lines = open(path).read().split("\n")
name, cnt, rows = None, 0, []
for l in lines:
    m = re.match(r"^(_ZN3nwe18render_mfma_kernel\S*):", l)
    if m:
        name, cnt = m.group(1), 0
        continue
    if name is None or want not in name:
        continue
    if "s_endpgm" in l:
        print(name, "mfma total", cnt, "scratch ops", len(rows))
        for r in rows:
            print("  after mfma", r[0], r[1])
        name, rows = None, []
        continue
    if "v_mfma" in l:
        cnt += 1
    if "scratch_" in l:
        rows.append((cnt, l.strip()[:70]))
