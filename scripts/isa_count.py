"""Counts instructions in the main loop of one fused-kernel instantiation (dev tool).
usage: python scripts/isa_count.py [/tmp/t/fused.s] [mangled-substring] [--json]
--json: one line {"valu": n, "cheap": n, "cheap_class_share": x} -- the share of the loop's vector instructions that belong
to the cheap issue class of profiles/r03_issue_rate_ubench.log (2.3-2.6 cycles in a stream of their own kind): add / sub /
and / or / xor / not / lshr / ashr / mov / cndmask / bitop3 / add_u16 / f32 add-mul-fma in their plain VOP forms with
vector-register or inline-constant sources (an SGPR source, DPP or SDWA makes them full-rate class)."""
import collections
import re
import sys

as_json = "--json" in sys.argv
argv = [a for a in sys.argv if a != "--json"]
path = argv[1] if len(argv) > 1 else "/tmp/t/fused.s"
key = argv[2] if len(argv) > 2 else "k_fused_chainILi5ELb0ELi1E"
CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32",
         "v_ashrrev_i32", "v_mov_b32", "v_cndmask_b32", "v_bitop3_b32", "v_add_u16", "v_fma_f32", "v_fmac_f32", "v_mul_f32",
         "v_add_f32"}
s = open(path).read()
m = re.search(r"^(_ZN3ipf\w*%s\w*):" % key, s, re.M)
i = m.start()
j = s.index(".Lfunc_end", i)
body = s[i:j].split("\n")
labels = {l.split(":")[0]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
loops = []
for k, l in enumerate(body):
    mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
    if mm:
        t = mm.group(1) or mm.group(2)
        if t in labels and labels[t] < k:
            loops.append((labels[t], k))
a, b = max(loops, key=lambda x: x[1] - x[0])
cnt = collections.Counter()
cheap = 0
for l in body[a:b]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    cnt[op] += 1
    base = re.sub(r"_e(32|64)$", "", op)
    if base in CHEAP:
        ops = l.split(";")[0].split(None, 1)[1] if " " in l else ""
        srcs = ops.split(",")[1:]
        if base == "v_cndmask_b32":
            srcs = srcs[:2]  # the lane mask (vcc / an SGPR pair) is not a data source
        if not any(re.match(r"\s*(s\d+|s\[|vcc|exec|ttmp)", x) for x in srcs):
            cheap += 1
tot = sum(cnt.values())
v = sum(c for o, c in cnt.items() if o.startswith("v_"))
sal = sum(c for o, c in cnt.items() if o.startswith("s_"))
if as_json:
    import json
    print(json.dumps({"kernel": m.group(1), "valu": v, "cheap": cheap, "cheap_class_share": cheap / v, "valu_per_32px": v}))
    sys.exit(0)
print("cheap-class", cheap, "of", v, "valu = %.3f" % (cheap / v))
print(m.group(1), "loop lines", b - a, "total", tot, "valu", v, "salu", sal, "valu/px", v / 32.0)
for o, c in cnt.most_common(60):
    print(f"{o:28s}{c}")
