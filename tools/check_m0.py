"""The MFMA render kernels keep the LDS-DMA destination in M0 across statements (one write per group of pieces), which is
sound only while hipcc emits no M0 use of its own in those kernels: disassemble the built library and check that every
instruction touching m0 in a render_mfma_kernel is one of ours - `s_mov_b32 m0, <scalar register>` immediately followed by
`s_nop 0` and the `global_load_lds_dwordx4` it addresses, the one asm statement of Walker::piece - and that no instruction
with an IMPLICIT M0 operand (s_set_gpr_idx_*, movrel, GWS/GDS, sendmsg, buffer loads with lds, ...) appears at all.

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
        pieces = 0
        # instructions that read or write M0 WITHOUT spelling it (gfx9 ISA): dynamic VGPR indexing (s_set_gpr_idx_* writes
        # M0[7:0] and M0[15:12]), relative moves, GWS / GDS / ordered-count, message sends, trace data, interpolation, and
        # any buffer load with the lds bit.  None has a reason to appear in these kernels; any of them voids the invariant.
        implicit = re.compile(r"^(s_set_gpr_idx_\w+|s_movrel\w*|v_movrel\w*|ds_gws_\w+|ds_ordered_count|ds_\w+_gs\w*|s_sendmsg\w*|s_ttracedata\w*|"
                              r"v_interp_\w+|s_getreg_b32 \S+ hwreg\(HW_REG_M0|buffer_\w+ .*\blds\b|ds_\w+ .*\bgds\b)")
        for name in sorted(os.listdir(tmp)):
            if not name.endswith("gfx950"):
                continue
            dis = subprocess.run([OBJDUMP, "-d", name], cwd=tmp, check=True, capture_output=True, text=True).stdout
            in_kernel = False
            body = []           # instructions of the current render_mfma_kernel, in order

            def close():
                nonlocal checked, pieces
                for i, ins in enumerate(body):
                    if implicit.match(ins):
                        raise AssertionError(f"instruction with an implicit M0 operand in the MFMA kernel: {ins!r}")
                    if ins.startswith("global_load_lds_dwordx4"):
                        pieces += 1
                    if re.search(r"\bm0\b", ins):
                        if not re.fullmatch(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)", ins):
                            raise AssertionError(f"unexpected M0 use in the MFMA kernel: {ins!r}")
                        # ours come as ONE asm statement: the write, one wait state, the piece it addresses (Walker::piece);
                        # a compiler-emitted write of the same spelling would not be followed by exactly that
                        nxt = body[i + 1:i + 3]
                        if len(nxt) < 2 or nxt[0] != "s_nop 0" or not nxt[1].startswith("global_load_lds_dwordx4"):
                            raise AssertionError(f"M0 write not followed by `s_nop 0` + LDS-DMA piece (not one of ours?): {[ins] + nxt!r}")
                        checked += 1
                body.clear()

            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    close()
                    in_kernel = "render_mfma_kernel" in m.group(1)
                    continue
                if not in_kernel:
                    continue
                ins = line.split("//")[0].strip()
                if ins:
                    body.append(ins)
            close()
        if checked > pieces:
            raise AssertionError(f"{checked} M0 writes but only {pieces} LDS-DMA pieces")
    if checked == 0:
        raise AssertionError("no LDS-DMA destination writes found: is this the right code object?")
    return checked


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                 "nerf-workspaces-explorer_amd", "libnwe_hip.so")
    print("M0 writes checked:", check(path))
