"""The MFMA render kernels keep the LDS-DMA destination in M0 across statements (one write per group of pieces), which is
sound only while hipcc emits no M0 use of its own in those kernels: disassemble the built library and check, for every
render_mfma_kernel, that

* every instruction touching m0 is one of ours: `s_mov_b32 m0, <scalar register>` immediately followed by `s_nop 0` and the
  `global_load_lds_dwordx4` it addresses - the one asm statement of Walker::piece;
* no instruction with an IMPLICIT M0 operand (s_set_gpr_idx_*, movrel, GWS/GDS, sendmsg, buffer loads with lds, ...) appears;
* along straight-line code the pieces between two M0 writes sit at consecutive 1-KiB instruction offsets and a group starts at
  offset 0 (a tail at 2048 or 3072): a piece at the wrong offset writes another piece's tile or past the LDS allocation
  (round 3 built exactly that once, in an experiment that moved the M0 write ahead of its piece, and faulted a box with it).

Run by `__graft_entry__.build()` (a build that breaks the invariant fails) and by tests/test_abi.py; `analyse()` is tested on
hand-made instruction lists in tests/test_host_logic.py."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# instructions that read or write M0 WITHOUT spelling it (gfx9 ISA): dynamic VGPR indexing (s_set_gpr_idx_* writes M0[7:0] and
# M0[15:12]), relative moves, GWS / GDS / ordered-count, message sends, trace data, interpolation, and any buffer load with
# the lds bit.  None has a reason to appear in these kernels; any of them voids the invariant.
IMPLICIT = re.compile(r"^(s_set_gpr_idx_\w+|s_movrel\w*|v_movrel\w*|ds_gws_\w+|ds_ordered_count|ds_\w+_gs\w*|s_sendmsg\w*|s_ttracedata\w*|"
                      r"v_interp_\w+|s_getreg_b32 \S+ hwreg\(HW_REG_M0|buffer_\w+ .*\blds\b|ds_\w+ .*\bgds\b)")
PIECE = "global_load_lds_dwordx4"


def analyse(body, addrs=None, x3=True):
    """(M0 writes, pieces) of one kernel's instruction list; raises AssertionError on a violation.  `addrs`: byte address per
    instruction (for branch targets), or None.  x3 = the three-product instantiation: the single-pass kernels skip the lo
    pieces (every second one) on purpose, so the continuity rule is theirs alone."""
    addrs = addrs or [None] * len(body)
    targets = set()
    for ins, addr in zip(body, addrs):       # join points: the continuity rule holds along straight-line code only
        mb = re.match(r"s_c?branch\w*\s+(\d+)$", ins)
        if mb and addr is not None:
            rel = int(mb.group(1))
            rel = rel - 65536 if rel >= 32768 else rel
            targets.add(addr + 4 + 4 * rel)
    last_off, unknown = None, True
    for ins, addr in (zip(body, addrs) if x3 else []):
        if (addr is not None and addr in targets) or ins.startswith(("s_cbranch", "s_branch")):
            last_off, unknown = None, True   # a piece behind a join may continue a group opened on either path
            continue
        if re.search(r"\bm0\b", ins):
            last_off, unknown = None, False
        elif ins.startswith(PIECE):
            mo = re.search(r"offset:(\d+)", ins)
            off = int(mo.group(1)) if mo else 0
            if last_off is not None and off != last_off + 1024:
                raise AssertionError(f"LDS-DMA piece at offset {off} behind one at {last_off} with no M0 write in between")
            if last_off is None and not unknown and off not in (0, 2048, 3072):
                raise AssertionError(f"first piece behind an M0 write at offset {off} (a group starts at 0, a tail at 2048 or 3072)")
            last_off, unknown = off, False
    writes = pieces = 0
    for i, ins in enumerate(body):
        if IMPLICIT.match(ins):
            raise AssertionError(f"instruction with an implicit M0 operand in the MFMA kernel: {ins!r}")
        if ins.startswith(PIECE):
            pieces += 1
        if re.search(r"\bm0\b", ins):
            if not re.fullmatch(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)", ins):
                raise AssertionError(f"unexpected M0 use in the MFMA kernel: {ins!r}")
            # ours come as ONE asm statement: the write, one wait state, the piece it addresses (Walker::piece); a
            # compiler-emitted write of the same spelling would not be followed by exactly that
            nxt = body[i + 1:i + 3]
            if len(nxt) < 2 or nxt[0] != "s_nop 0" or not nxt[1].startswith(PIECE):
                raise AssertionError(f"M0 write not followed by `s_nop 0` + LDS-DMA piece (not one of ours?): {[ins] + nxt!r}")
            writes += 1
    if writes > pieces:
        raise AssertionError(f"{writes} M0 writes but only {pieces} LDS-DMA pieces")
    return writes, pieces


def check(lib_path: str) -> int:
    """Returns the number of M0 writes found (all of the permitted form); raises on a violation or if nothing could be
    checked."""
    if not os.path.exists(OBJDUMP):
        raise RuntimeError(f"{OBJDUMP} is missing: the kernel-owned-M0 invariant cannot be checked")
    checked = 0
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(lib_path, os.path.join(tmp, "libnwe_hip.so"))
        subprocess.run([OBJDUMP, "--offloading", "libnwe_hip.so"], cwd=tmp, check=True, capture_output=True)
        for name in sorted(os.listdir(tmp)):
            if not name.endswith("gfx950"):
                continue
            dis = subprocess.run([OBJDUMP, "-d", name], cwd=tmp, check=True, capture_output=True, text=True).stdout
            in_kernel, x3, body, addrs = False, False, [], []
            for line in dis.splitlines() + ["0 <end>:"]:
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    if body:
                        checked += analyse(body, addrs, x3)[0]
                    body, addrs = [], []
                    in_kernel = "render_mfma_kernel" in m.group(1)
                    mx = re.search(r"render_mfma_kernelILi\d+ELi\d+ELi(?:n?\d+)ELb([01])", m.group(1))
                    x3 = bool(mx and mx.group(1) == "1")
                    continue
                if not in_kernel:
                    continue
                ins = line.split("//")[0].strip()
                if ins:
                    body.append(ins)
                    ma = re.search(r"//\s*([0-9A-Fa-f]+):", line)
                    addrs.append(int(ma.group(1), 16) if ma else None)
    if checked == 0:
        raise AssertionError("no LDS-DMA destination writes found: is this the right code object?")
    return checked


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                 "nerf-workspaces-explorer_amd", "libnwe_hip.so")
    print("M0 writes checked:", check(path))
