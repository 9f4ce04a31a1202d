"""Instruction classes of a kernel's largest innermost loop (tools, not product): convert / multiply-add / lane-crossing (DPP) /
maxima / integer sums / masks and bit operations / address and other integer / LDS / memory / scalar, from the assembly
that `hipcc -S` writes.  Usage: isa_classes.py file.s <mangled name prefix> [samples per lane and trip]"""
import re, sys, collections
s = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(s) if l.startswith(sys.argv[2]))
end = next(i for i in range(start, len(s)) if "s_endpgm" in s[i])
body = s[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB[0-9_]+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB[0-9_]+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
# the largest loop that holds no other loop (the main loop of the kernels here; run-level loops hold it)
inner = [a for a in loops if not any(b != a and a[0] <= b[0] and b[1] <= a[1] for b in loops)]
best = max(inner, key=lambda a: a[1] - a[0])
def cls(op):
    if op.endswith("_dpp"): return "lane-crossing (DPP)"
    if op.startswith(("v_cvt", "v_bfe")): return "convert"
    if op.startswith(("v_fma", "v_fmac", "v_fmamk", "v_fmaak", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_", "v_trunc", "v_sqrt", "v_rcp")): return "multiply-add / float"
    if op.startswith(("v_max", "v_min", "v_med3")): return "maxima"
    if op.startswith(("v_dot2", "v_sdot")): return "integer sums (dot2)"
    if op.startswith(("v_and", "v_or", "v_xor", "v_bfi", "v_not", "v_cndmask", "v_alignbit", "v_perm", "v_cmp")): return "masks, selects, compares"
    if op.startswith("v_"): return "address / other integer / moves"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "memory"
    if op.startswith("s_"): return "scalar"
    return "other"
c = collections.Counter()
for l in body[best[0]:best[1] + 1]:
    m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
    if m: c[cls(m.group(1))] += 1
per = float(sys.argv[3]) if len(sys.argv) > 3 else None
valu = sum(v for k, v in c.items() if k not in ("LDS", "memory", "scalar", "other"))
print(f"largest innermost loop: lines {best[0]}..{best[1]}; VALU {valu}" + (f" = {valu / per:.2f} per sample and lane" if per else ""))
for k, v in c.most_common():
    print(f"  {k:34s} {v:5d}" + (f"  {v / per:5.2f} per sample" if per else ""))
