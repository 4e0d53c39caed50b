"""Instruction-slot view of the MFMA kernel's generated assembly (after `make asm`).

With one wave per SIMD every instruction costs one issue slot, and a v_mfma_f32_32x32x16_f16 hides about seven of them;
this prints, for the barrier-to-barrier stretch of a few tiles, the number of non-MFMA instructions between consecutive MFMAs.
"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "nerf-workspaces-explorer_amd/csrc/build/nwe_kernel_mfma.s"
tiles = [int(t) for t in sys.argv[2:]] or [14, 20]
s = open(path).read()
L = s[: s.index(".end_amdhsa_kernel")].split("\n")
bars = [i for i, l in enumerate(L) if re.match(r"\s*s_barrier", l)]
for k in tiles:
    seg = [l.strip() for l in L[bars[k]:bars[k + 1]] if l.strip() and not l.strip().startswith((";", "."))]
    gaps, g = [], 0
    for l in seg:
        if l.startswith("v_mfma"):
            gaps.append(g)
            g = 0
        else:
            g += 1
    gaps.append(g)
    n_mfma = len(gaps) - 1
    slots = sum(max(8, x + 1) for x in gaps[1:]) if n_mfma else 0
    print(f"tile {k}: {len(seg)} instr, {n_mfma} mfma, {(len(seg) - n_mfma) / max(n_mfma, 1):.2f} other/mfma, "
          f"slot model {slots / (8 * max(n_mfma, 1)):.2f}x mfma-bound; gaps {gaps}")
