#!/usr/bin/env python3
"""Kernel resource table of the built library, read from the device assembly the build keeps
(plonky2-bls12-381-pairing_amd/build/*/*gfx950.s): registers, private segment (scratch), LDS, occupancy, code size and
instruction statistics per kernel.  DESIGN.md 3.4's table is this script's output:
    python tools/kernel_table.py            markdown table
    python tools/kernel_table.py --json     the same as JSON (tests/test_build_guard.py checks DESIGN.md's figures against it)
"""
import glob, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "build")
INSTR = re.compile(r"^\s+([vs]_\w+|ds_\w+|global_\w+|scratch_\w+|buffer_\w+|flat_\w+)")


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"\(.*", "", o.replace("void ", "")) for o in out[: len(names)]]
    except Exception:
        return names


def kernels():
    rows = []
    for path in sorted(glob.glob(os.path.join(BUILD, "*", "*gfx950.s"))):
        unit = os.path.basename(os.path.dirname(path))
        lines = open(path).read().split("\n")
        starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^\s+\.amdhsa_kernel (\S+)", l)] if m]
        for i, name in starts:
            meta = {}
            for l in lines[i: i + 80]:
                m = re.match(r"^\s+\.amdhsa_(group_segment_fixed_size|private_segment_fixed_size|next_free_vgpr|accum_offset) (\d+)", l)
                if m:
                    meta[m.group(1)] = int(m.group(2))
            info = {}
            for l in lines[i: i + 140]:
                m = re.match(r"^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|codeLenInByte)\s*[:=]\s*(\d+)", l)
                if m and m.group(1) not in info:
                    info[m.group(1)] = int(m.group(2))
            # body: from the kernel's label to its s_endpgm-terminated end (next function label)
            lab = next((j for j, l in enumerate(lines) if l.startswith(name + ":")), None)
            cnt, mad, scr, acc = 0, 0, 0, 0
            if lab is not None:
                for l in lines[lab + 1:]:
                    if re.match(r"^\s+\.section|^\.Lfunc_end", l):
                        break
                    m = INSTR.match(l)
                    if m:
                        cnt += 1
                        op = m.group(1)
                        mad += op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32")
                        scr += op.startswith("scratch_")
                        acc += op.startswith("v_accvgpr")
            rows.append({"unit": unit, "symbol": name, "vgpr": info.get("NumVgprs"), "agpr": info.get("NumAgprs"),
                         "regs": info.get("TotalNumVgprs", meta.get("next_free_vgpr")), "scratch_bytes": meta.get("private_segment_fixed_size"),
                         "lds_bytes": meta.get("group_segment_fixed_size"), "occupancy": info.get("Occupancy"),
                         "code_bytes": info.get("codeLenInByte"), "static_instr": cnt, "static_mad": mad, "static_scratch_ops": scr,
                         "static_accvgpr_moves": acc})
    for r, d in zip(rows, demangle([r["symbol"] for r in rows])):
        r["kernel"] = d
    return rows


if __name__ == "__main__":
    rows = kernels()
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1))
    else:
        print("| kernel (translation unit) | registers (VGPR + AGPR) | private segment | LDS | waves / SIMD | code | static instr. (multiply-adds, scratch ops, AGPR moves) |")
        print("|---|---|---|---|---|---|---|")
        for r in rows:
            print("| `%s` (%s) | %s (%s + %s) | %s B | %s B | %s | %.0f KB | %d (%d, %d, %d) |" % (
                r["kernel"], r["unit"], r["regs"], r["vgpr"], r["agpr"], r["scratch_bytes"], r["lds_bytes"], r["occupancy"],
                (r["code_bytes"] or 0) / 1024, r["static_instr"], r["static_mad"], r["static_scratch_ops"], r["static_accvgpr_moves"]))
