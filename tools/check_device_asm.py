#!/usr/bin/env python3
"""Build guard for a hipcc (ROCm 7.2 / LLVM 22) code-generation hazard seen on gfx950:

when a branch inside a device FUNCTION (not a kernel) is out of the +-128 KB range of s_branch, branch relaxation
materialises the target with `s_getpc_b64 s[30:31] ... s_setpc_b64 s[30:31]`.  In a LEAF function s[30:31] still
holds the return address (nothing spilled it), so the function returns into its own body and the wave never ends --
a GPU hang, not a wrong answer.  Kernels are unaffected (no return address); non-leaf functions are unaffected (the
prologue saves s[30:31] with v_writelane and the epilogue restores it).

usage: check_device_asm.py file.s [...]   -> exit 1 and a report if any leaf device function uses the pattern.
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
    sys.exit(1 if failed else 0)
