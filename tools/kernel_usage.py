"""Register / scratch / LDS use of every render_mfma_kernel instantiation, from `make -C csrc asm`."""
import re, sys
t = open(sys.argv[1] if len(sys.argv) > 1 else "nerf-workspaces-explorer_amd/csrc/build/nwe_kernel_mfma.usage.txt").read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0]
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    m = re.search(r"render_mfma_kernelILi(\d+)ELi(\d+)ELi(-?\w+)ELb(\d)ELb(\d)ELi(\d)ELb(\d)", name)
    if m:
        W, D, S, x3, sp, fo, lean = m.groups()
        scratch, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]")
        print(f"W={W} D={D} skip={S.replace('n', '-')} X3={x3} SPLIT={sp} FORM={fo} LEAN={lean}  VGPR {g('VGPRs')} AGPR {g('AGPRs')} SGPR {g('TotalSGPRs')} "
              f"scratch {scratch} LDS {lds}")
