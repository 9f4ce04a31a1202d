"""Issue-cycle estimate of a kernel's main loop from its assembly (tools, not product).

Measured on gfx950 (tools/micro/pk_rate.hip): a wavefront's plain f32 / 32-bit integer VOP1/VOP2 ops
and v_fma_f32 with register or literal operands take about half the SIMD time of the rest (max, cvt,
bfe, alignbit, dot2, DPP, SDWA, packed f32, 3-operand integer ops, SGPR operands); sqrt / rcp about 3x.
Usage: isa_cycles.py file.s <substring of the kernel's mangled name> [loop index]"""
import re, sys
from collections import Counter
FAST = ("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_lshlrev_b32",
        "v_lshrrev_b32", "v_mov_b32", "v_not_b32")
TRANS = ("v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32")
def weight(op):
    base = re.sub(r"_e32$|_e64$", "", op)
    if base.endswith(("_dpp", "_sdwa")): return 4
    if base in TRANS: return 8
    if base in FAST: return 2
    return 4 if base.startswith("v_") else 0
s = open(sys.argv[1]).read()
for nm in re.findall(r"\n(_Z\w+):", s):
    if sys.argv[2] not in nm: continue
    i = s.index("\n" + nm + ":"); j = s.index(".Lfunc_end", i)
    body = s[i:j]
    loops = [m.start() for m in re.finditer(r"Loop Header", body)]
    if not loops: continue
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    a = loops[k]; b = body.find("s_branch", a)
    ins = [l.split()[0] for l in body[a:b].split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = Counter(ins)
    cyc = sum(weight(o) * n for o, n in c.items())
    fast = sum(n for o, n in c.items() if weight(o) == 2); slow = sum(n for o, n in c.items() if weight(o) == 4)
    print(f"{nm[:70]}: {len(ins)} instr, VALU fast {fast} slow {slow} trans {sum(n for o, n in c.items() if weight(o) == 8)} -> {cyc} cycles")
    print("   slow:", {o: n for o, n in c.most_common() if weight(o) >= 4})
