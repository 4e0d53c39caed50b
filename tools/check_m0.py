"""The MFMA render kernels keep the LDS-DMA destination in M0 across statements (one write per group of pieces), which is
sound only while hipcc emits no M0 use of its own in those kernels: disassemble the built library and check that every
instruction touching m0 in a render_mfma_kernel is one of ours (`s_mov_b32 m0, <scalar register>`).

Run by `__graft_entry__.build()` (a build that breaks the invariant fails) and by tests/test_abi.py."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def check(lib_path: str) -> int:
    """Returns the number of M0 writes found (all of the permitted form); raises on a foreign M0 use or if nothing could be
    checked."""
    if not os.path.exists(OBJDUMP):
        raise RuntimeError(f"{OBJDUMP} is missing: the kernel-owned-M0 invariant cannot be checked")
    with tempfile.TemporaryDirectory() as tmp:
        lib = os.path.join(tmp, "libnwe_hip.so")
        shutil.copy(lib_path, lib)
        subprocess.run([OBJDUMP, "--offloading", "libnwe_hip.so"], cwd=tmp, check=True, capture_output=True)
        checked = 0
        for name in sorted(os.listdir(tmp)):
            if not name.endswith("gfx950"):
                continue
            dis = subprocess.run([OBJDUMP, "-d", name], cwd=tmp, check=True, capture_output=True, text=True).stdout
            in_kernel = False
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    in_kernel = "render_mfma_kernel" in m.group(1)
                    continue
                if not in_kernel:
                    continue
                ins = line.split("//")[0].strip()
                if re.search(r"\bm0\b", ins):
                    if not re.fullmatch(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)", ins):
                        raise AssertionError(f"unexpected M0 use in the MFMA kernel: {ins!r}")
                    checked += 1
    if checked == 0:
        raise AssertionError("no LDS-DMA destination writes found: is this the right code object?")
    return checked


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                 "nerf-workspaces-explorer_amd", "libnwe_hip.so")
    print("M0 writes checked:", check(path))
