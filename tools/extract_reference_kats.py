#!/usr/bin/env python3
"""Collect the numeric known answers the reference's own files hold for this path (SURVEY.md section 4,
KAT-1..KAT-6) into tests/golden/reference_kats.json.  Only NUMBERS are extracted (6-limb u64 groups inside
`Bls12_381Base([...])` literals, the p digits, the loop constants) -- no source text is copied.

Run here (needs /root/reference, which does not exist on the GPU box); the JSON it writes is committed.
"""
import json, os, re, sys

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_kats.json")
LIMB = re.compile(r"0x[0-9a-fA-F_]+")


def limb_groups(path, lo, hi):
    """All Bls12_381Base([..6 limbs..]) literals between lines lo..hi (1-based, inclusive) as ints (LE limbs)."""
    lines = open(os.path.join(REF, path)).read().split("\n")[lo - 1:hi]
    text = "\n".join(lines)
    groups = []
    for m in re.finditer(r"Bls12_381Base\(\[(.*?)\]\)", text, re.S):
        limbs = [int(x.replace("_", ""), 16) for x in LIMB.findall(m.group(1))]
        assert len(limbs) == 6, (path, lo, limbs)
        groups.append(sum(l << (64 * i) for i, l in enumerate(limbs)))
    return groups


k = {}
g = limb_groups("fields_as_trees/fq2_target_tree.rs", 221, 300)
k["kat1_fp2_add"] = {"src": "src/fields_as_trees/fq2_target_tree.rs:225-298", "a": g[0:2], "b": g[2:4], "c": g[4:6]}
g = limb_groups("fields_as_trees/fq2_target_tree.rs", 330, 415)
k["kat2_fp2_sub"] = {"src": "src/fields_as_trees/fq2_target_tree.rs:340-411", "a": g[0:2], "b": g[2:4], "c": g[4:6]}
g1 = limb_groups("curves/g1_curve_target.rs", 36, 64)
g2 = limb_groups("curves/g2_curve_target.rs", 62, 116)
k["kat3_generators_montgomery"] = {"src": "src/curves/g1_curve_target.rs:42-59, src/curves/g2_curve_target.rs:69-112",
                                   "g1_x": g1[0], "g1_y": g1[1], "g2_x": g2[0:2], "g2_y": g2[2:4]}
f6 = limb_groups("fields_as_trees/fq6_target_tree.rs", 129, 169)
f12 = limb_groups("fields_as_trees/fq12_target_tree.rs", 92, 128)
k["kat4_frobenius_montgomery"] = {"src": "src/fields_as_trees/fq6_target_tree.rs:139-162, fq12_target_tree.rs:101-119",
                                  "frob6_c1_c1": f6[0], "frob6_c2_c0": f6[1], "frob12_c1": f12[0:2]}
# KAT-5: the fixed limb inputs a, b, c of test_fq6_arithmetic / test_fq12_arithmetic (identities x^2 == x*x and
# (a+b) c^2 == c^2 a + c^2 b; no expected outputs are stored in the reference)
g = limb_groups("fields_as_trees/fq6_target_tree.rs", 392, 625)
assert len(g) == 18, len(g)
k["kat5_fq6_inputs"] = {"src": "src/fields_as_trees/fq6_target_tree.rs:392-620", "a": g[0:6], "b": g[6:12], "c": g[12:18]}
g = limb_groups("fields_as_trees/fq12_target_tree.rs", 448, 925)
assert len(g) == 36, len(g)
k["kat5_fq12_inputs"] = {"src": "src/fields_as_trees/fq12_target_tree.rs:448-920", "a": g[0:12], "b": g[12:24], "c": g[24:36]}
# KAT-6: p as u32 digits, loop constants
txt = open(os.path.join(REF, "fields/bls12_381base.rs")).read().split("\n")[107:113]
digits = [int(x, 16) for x in re.findall(r"0x[0-9A-Fa-f]{8}", "\n".join(txt))]
assert len(digits) == 12
k["kat6_p_u32_digits"] = {"src": "src/fields/bls12_381base.rs:108-113", "digits_le": digits}
txt = open(os.path.join(REF, "utils/constants.rs")).read()
k["kat6_bls_x"] = {"src": "src/utils/constants.rs:1-2", "bls_x": int(re.search(r"0x[0-9a-f_]+", txt).group(0).replace("_", ""), 16),
                   "negative": "true" in txt}
txt = open(os.path.join(REF, "global_constants.rs")).read()
body = txt[txt.index("PSEUDO_BINARY_ENCODING"):]
bits = [int(x) for x in re.findall(r"\b[01]\b", body[body.index("=") :body.index("];")])]
k["kat6_pseudo_binary_encoding"] = {"src": "src/global_constants.rs:3-6", "bits": bits}
txt = open(os.path.join(REF, "miller_loop_native_optimized.rs")).read()
k["kat6_scalar_field"] = {"src": "src/miller_loop_native_optimized.rs:104,110",
                          "p_decimal": re.search(r'base_field_str = "(\d+)"', txt).group(1),
                          "r_decimal": re.search(r'b"(\d+)"', txt).group(1)}
json.dump(k, open(OUT, "w"), indent=1, default=str)
print("wrote", os.path.abspath(OUT), {n: (len(v) if isinstance(v, (list, dict)) else v) for n, v in k.items()})
