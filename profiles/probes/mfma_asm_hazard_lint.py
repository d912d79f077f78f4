#!/usr/bin/env python
"""ISA lint for the hand-pinned MFMAs of csrc/encoder.hip.

encoder.hip issues its main-loop MFMAs as `asm volatile("v_mfma_f32_32x32x16_f16 ...")` statements so that the hand-written interleave
of MFMAs, LDS reads and global loads survives the scheduler (encoder.hip: mfma16_pinned).  The price: the compiler's hazard recogniser
does not know that such a statement is an MFMA, so it inserts NO wait states between it and a later non-MFMA instruction that touches
the accumulator (gfx950: an 8-pass XDL write needs ~11 wait states before a VALU / v_accvgpr_read / v_accvgpr_write access to the same
registers) — the source keeps that distance by construction (mfma_fence / mfma_results_ready), and a register-allocator copy of an
accumulator between two pinned MFMAs would silently break it.  This lint checks the BUILT code object:

    for every v_mfma_* instruction, no non-MFMA instruction reads or writes a register of its destination within the next
    REQUIRED wait states (s_nop N counts N + 1, every other instruction 1 — conservative: MFMAs and memory instructions take longer)

along the straight-line order of every kernel and, for every backward branch, along the path loop end -> loop head.
MFMA -> MFMA on the same accumulator is interlocked by the hardware and is not checked.

Rule B (the other direction): no VALU instruction may write a register that an MFMA reads (SrcA / SrcB / SrcC) within the
REQUIRED_RAW wait states before that MFMA.  Found on the GPU: in the 256-register two-waves-per-SIMD instantiation the compiler
reloads spilled B fragments with v_accvgpr_read right in front of the pinned MFMA that consumes them; with the reload as the
instruction directly before the MFMA the features were wrong (deterministically), one instruction earlier they were right, and
`-mllvm -amdgpu-snop-padding=1` cured every failing variant (profiles/NOTES.md section O).  For a builtin MFMA the compiler keeps
>= 4 (encoder_general.hip.o); the pinned MFMAs of that instantiation therefore carry `s_nop 1` inside their statement.

Rule C (`--dpp`, for the Kuka kernels): the lane-group primitives are DPP instructions inside asm statements (csrc/kuka_group.hpp
fmac_bcast, the projected Gauss-Seidel rows of kuka_tree.hpp); a DPP read of a VGPR needs 2 wait states after a VALU write of it, which
the statements provide themselves (`s_nop 1` / an independent instruction in between).  Checked on the built object: for every
*_dpp instruction, the closest preceding VALU write of its DPP source operand.

    python profiles/probes/mfma_asm_hazard_lint.py [object=robotics-rl-srl_amd/csrc/build/encoder.hip.o] [kernel-regex=.]
    python profiles/probes/mfma_asm_hazard_lint.py --dpp [object=robotics-rl-srl_amd/csrc/build/kuka_tree.hip.o]
exit status 1 and one line per violation if any; used by tests/test_isa_lint.py."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
# Wait states between an 8-pass XDL write and a VALU access of the result.  Calibrated on the compiler's own code in the same object
# (the builtin MFMAs of the two-waves-per-SIMD instantiation): the closest v_accvgpr_read the hazard recogniser allows there has 12
# instructions / nop cycles between itself and the MFMA — the lint reports nothing at 12 and the compiler's own code at 13.  The
# hand-pinned streams keep >= 17.
REQUIRED = 12
REQUIRED_RAW = 2       # rule B: VALU write of an operand register -> MFMA (0 fails on the GPU, 1 was observed to work)
WINDOW = 64            # instructions followed past a back-edge

_reg = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def disassemble(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
        targets = subprocess.check_output([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + fat], text=True).split()
        t = [x for x in targets if "gfx950" in x][0]
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=" + t, "--output=" + co],
                              stderr=subprocess.DEVNULL)
        return subprocess.check_output([LLVM + "/llvm-objdump", "-d", co], text=True)


def regs_of(text):
    out = set()
    for m in _reg.finditer(text):
        if m.group(4) is not None:
            out.add((m.group(1), int(m.group(4))))
        else:
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class Ins:
    __slots__ = ("addr", "mnem", "ops", "is_mfma", "dst", "regs", "ws", "target", "wdst")

    def __init__(self, line):
        body, _, tail = line.partition("//")
        parts = body.strip().split(None, 1)
        self.mnem = parts[0]
        self.ops = parts[1] if len(parts) > 1 else ""
        m = re.match(r"\s*([0-9A-Fa-f]+):\s*([0-9A-Fa-f]{8})", tail)
        self.addr = int(m.group(1), 16) if m else None
        self.is_mfma = self.mnem.startswith("v_mfma") or self.mnem.startswith("v_smfmac")
        self.regs = regs_of(self.ops)
        self.dst = regs_of(self.ops.split(",")[0]) if self.is_mfma else set()
        self.wdst = regs_of(self.ops.split(",")[0]) if (self.mnem.startswith("v_") and not self.is_mfma) else set()
        self.ws = int(self.ops.strip()) + 1 if self.mnem == "s_nop" else 1
        self.target = None
        if self.mnem.startswith("s_cbranch") or self.mnem == "s_branch":
            simm = int(m.group(2), 16) & 0xFFFF
            if simm >= 0x8000:
                simm -= 0x10000
            self.target = self.addr + 4 + 4 * simm


def kernels(asm):
    """{demangled name: [Ins]}"""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = out.setdefault(name, [])
            continue
        if cur is not None and line.startswith("\t") and "//" in line:
            try:
                cur.append(Ins(line))
            except Exception:
                pass
    return out


def scan(seq, where, found):
    pending = {}                              # register -> (wait states since the MFMA that wrote it, that MFMA)
    recent = []                               # rule B: (VALU instruction, wait states issued since it) of the last few wait states
    for ins in seq:
        if ins.is_mfma:
            ops = [o.strip() for o in ins.ops.split(",")]
            srcs = set().union(*[regs_of(o) for o in ops[1:4]])
            for pv, ws in recent:
                if ws < REQUIRED_RAW and pv.wdst & srcs:
                    r = sorted(pv.wdst & srcs)[0]
                    found.add("%s: %s %s at 0x%x writes %s%d only %d wait states before %s at 0x%x reads it" % (
                        where, pv.mnem, pv.ops.strip(), pv.addr, r[0], r[1], ws, ins.mnem, ins.addr))
        recent = [(pv, ws + ins.ws) for pv, ws in recent if ws + ins.ws < REQUIRED_RAW]
        if not ins.is_mfma and ins.mnem.startswith("v_"):
            recent.append((ins, 0))
        if not ins.is_mfma:
            for r in ins.regs:
                if r in pending and pending[r][0] < REQUIRED:
                    found.add("%s: %s %s at 0x%x touches %s%d only %d wait states after %s at 0x%x" % (
                        where, ins.mnem, ins.ops.strip(), ins.addr, r[0], r[1], pending[r][0], pending[r][1].mnem, pending[r][1].addr))
        for r in list(pending):
            c = pending[r][0] + ins.ws
            if c >= REQUIRED:
                del pending[r]
            else:
                pending[r] = (c, pending[r][1])
        if ins.is_mfma:
            for r in ins.dst:
                pending[r] = (0, ins)


def lint(obj, pattern="."):
    found, stats = set(), {}
    for name, seq in kernels(disassemble(obj)).items():
        if not re.search(pattern, name) or not any(i.is_mfma for i in seq):
            continue
        scan(seq, name, found)
        index = {i.addr: k for k, i in enumerate(seq)}
        back = 0
        for k, ins in enumerate(seq):
            if ins.target is not None and ins.target <= ins.addr and ins.target in index:
                back += 1
                head = index[ins.target]
                scan(seq[max(head, k - WINDOW):k + 1] + seq[head:head + WINDOW], name + " (back-edge at 0x%x)" % ins.addr, found)
        stats[name] = {"instructions": len(seq), "mfma": sum(i.is_mfma for i in seq), "back_edges": back}
    return sorted(found), stats


REQUIRED_DPP = 2


def lint_dpp(obj):
    """(violations, {distance: count}, number of DPP instructions): distance = wait states between a VALU write of a DPP instruction's
    permuted source (src0) and that instruction, over every kernel of the object (straight-line order; distances >= 6 not recorded)."""
    found, hist, n_dpp = [], {}, 0
    for name, seq in kernels(disassemble(obj)).items():
        recent = []
        for ins in seq:
            if "_dpp" in ins.mnem:
                n_dpp += 1
                ops = [o.strip() for o in ins.ops.split(",")]
                src0 = regs_of(ops[1].split()[0]) if len(ops) > 1 else set()
                for pv, ws in recent:
                    if pv.wdst & src0:
                        hist[ws] = hist.get(ws, 0) + 1
                        if ws < REQUIRED_DPP:
                            found.append("%s: %s %s at 0x%x writes the DPP source of %s %s at 0x%x only %d wait states before it" % (
                                name[:80], pv.mnem, pv.ops.strip(), pv.addr, ins.mnem, ins.ops.strip(), ins.addr, ws))
            recent = [(pv, ws + ins.ws) for pv, ws in recent if ws + ins.ws < 6]
            if ins.mnem.startswith("v_"):
                recent.append((ins, 0))
    return found, hist, n_dpp


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--dpp":
        repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        obj = sys.argv[2] if len(sys.argv) > 2 else os.path.join(repo, "robotics-rl-srl_amd", "csrc", "build", "kuka_tree.hip.o")
        found, hist, n = lint_dpp(obj)
        print("%d DPP instructions; VALU write of the DPP source -> DPP instruction, wait states: %s" % (n, sorted(hist.items())))
        for f in found:
            print("HAZARD " + f)
        print("%d violation(s)" % len(found))
        sys.exit(1 if found else 0)
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(repo, "robotics-rl-srl_amd", "csrc", "build", "encoder.hip.o")
    found, stats = lint(obj, sys.argv[2] if len(sys.argv) > 2 else ".")
    for name, st in stats.items():
        print("%s: %d instructions, %d MFMAs, %d back-edges checked" % (name[:90], st["instructions"], st["mfma"], st["back_edges"]))
    for f in found:
        print("HAZARD " + f)
    print("%d violation(s)" % len(found))
    sys.exit(1 if found else 0)
