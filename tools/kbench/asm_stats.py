#!/usr/bin/env python3
"""Static instruction statistics of a kernel's device assembly, per basic-block region (developer tool):
   tools/kbench/asm_stats.py <file.s> [kernel-substring]
Prints registers / scratch and, for every region between labels that holds more than 500 instructions, the
instruction count, the share of v_mad_u64_u32 and the most frequent other opcodes."""
import collections, re, sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(path).read().split("\n")
funcs = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) or re.match(r"^k_\w+:", l)]
INSTR = re.compile(r"^\s+([vs]_\w+|ds_\w+|global_\w+|scratch_\w+|buffer_\w+|flat_\w+)")
for n, (start, name) in enumerate(funcs):
    if want not in name:
        continue
    end = funcs[n + 1][0] if n + 1 < len(funcs) else len(lines)
    body = lines[start:end]
    meta = [l.strip() for l in body if re.search(r"NumVgprs|NumAgprs|ScratchSize|Occupancy", l)]
    print(name, " ".join(m.lstrip("; ") for m in meta))
    cuts = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(body)]
    prev = 0
    total = collections.Counter()
    for c in cuts:
        h = collections.Counter()
        for l in body[prev:c]:
            m = INSTR.match(l)
            if m:
                h[m.group(1)] += 1
        total.update(h)
        t = sum(h.values())
        if t > 500:
            mad = h["v_mad_u64_u32"]
            rest = " ".join("%s=%d" % (k.replace("v_", "").replace("_e32", "").replace("_e64", ""), v) for k, v in h.most_common(12) if k != "v_mad_u64_u32")
            print("  lines %6d-%6d  instr %6d  mad %6d (%4.1f%%)  %s" % (prev, c, t, mad, 100.0 * mad / t, rest))
        prev = c
    t = sum(total.values())
    print("  total %d  mad %d (%.1f%%)" % (t, total["v_mad_u64_u32"], 100.0 * total["v_mad_u64_u32"] / max(t, 1)))
