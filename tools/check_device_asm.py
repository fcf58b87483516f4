#!/usr/bin/env python3
"""Build guard for a hipcc (ROCm 7.2 / LLVM 22) code-generation hazard seen on gfx950:

when a branch inside a device FUNCTION (not a kernel) is out of the +-128 KB range of s_branch, branch relaxation
materialises the target with `s_getpc_b64 s[30:31] ... s_setpc_b64 s[30:31]`.  In a LEAF function s[30:31] still
holds the return address (nothing spilled it), so the function returns into its own body and the wave never ends --
a GPU hang, not a wrong answer.  Kernels are unaffected (no return address); non-leaf functions are unaffected (the
prologue saves s[30:31] with v_writelane and the epilogue restores it).

Second hazard (hardware vs compiler, measured with tools/microbench/dpp_sub_probe.hip on an MI355X,
gpurun_out/dpp_sub_probe.txt): in their DPP encoding the INTEGER "rev" VOP2 opcodes apply the lane permutation to the
OTHER operand than LLVM assumes --

    v_subrev_u32_dpp d, a, b     LLVM: b - dpp(a)      gfx950: dpp(b) - a        (same for v_subrev_co_u32_dpp)
    v_lshlrev_b32_dpp d, s, v    LLVM: v << dpp(s)     gfx950: dpp(v) << s

(v_sub_u32_dpp, v_add_u32_dpp, v_mov_b32_dpp and the float v_subrev_f32_dpp behave as documented).  LLVM's DPP combiner
(GCNDPPCombine) creates v_subrev_u32_dpp when it folds  o = v_mov_b32_dpp(a); d = b - o,  which is exactly the shape of
the lane-pair Fp2 squaring (a - swap(a)): the even lane then receives swap(a) - a, a silently NEGATED coefficient.  The
build disables the combiner (-mllvm -amdgpu-dpp-combine=false, Makefile) and this check fails if any such instruction
still appears.

usage: check_device_asm.py file.s [...]   -> exit 1 and a report if any translation unit shows either pattern.
The Makefile runs it on the assembly of every kernel translation unit (-save-temps=obj).
"""
import re
import sys


def check(path):
    bad = []
    name, body, is_kernel = None, [], False
    kernels = set()
    txt = open(path, errors="ignore").read()
    for m in re.finditer(r"^\s*\.amdhsa_kernel\s+(\S+)", txt, re.M):
        kernels.add(m.group(1))
    for line in txt.split("\n"):
        m = re.match(r"^([A-Za-z_.$][\w.$]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L"):
            if name is not None:
                bad += judge(name, body, name in kernels)
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
    if name is not None:
        bad += judge(name, body, name in kernels)
    return bad


# integer VOP2 opcodes whose operand order is reversed, in DPP form (see the module docstring)
REV_DPP = re.compile(r"^\s*(v_(?:subrev|subbrev|lshlrev|lshrrev|ashrrev)_(?:co_)?[a-z]*\d+_dpp)\b", re.M)


def rev_dpp_uses(path):
    """(mnemonic, count) of reversed-operand integer DPP instructions in an assembly file"""
    found = {}
    for m in REV_DPP.finditer(open(path, errors="ignore").read()):
        if "_f16" in m.group(1) or "_f32" in m.group(1) or "_f64" in m.group(1):
            continue
        found[m.group(1)] = found.get(m.group(1), 0) + 1
    return sorted(found.items())


def judge(name, body, is_kernel):
    if is_kernel:
        return []
    text = "\n".join(body)
    if "s_getpc_b64 s[30:31]" not in text:
        return []
    # a long branch through s[30:31]: fine only if the return address was saved first
    saved = re.search(r"v_writelane_b32 v\d+, s30,", text) is not None
    returns = "s_setpc_b64 s[30:31]" in text
    return [] if (saved or not returns) else [name]


if __name__ == "__main__":
    failed = False
    for p in sys.argv[1:]:
        for fn in check(p):
            failed = True
            print(f"{p}: device function {fn} relaxes a long branch through s[30:31] without saving the return address")
        for op, cnt in rev_dpp_uses(p):
            failed = True
            print(f"{p}: {cnt} x {op}: gfx950 permutes the other operand than LLVM assumes (build with -mllvm -amdgpu-dpp-combine=false)")
    sys.exit(1 if failed else 0)
