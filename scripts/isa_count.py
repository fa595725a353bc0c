"""Counts instructions in the main loop of one fused-kernel instantiation (dev tool).
usage: python scripts/isa_count.py [/tmp/t/fused.s] [mangled-substring]"""
import collections
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/t/fused.s"
key = sys.argv[2] if len(sys.argv) > 2 else "k_fused_chainILi5ELb0ELi1E"
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
for l in body[a:b]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    cnt[l.split()[0]] += 1
tot = sum(cnt.values())
v = sum(c for o, c in cnt.items() if o.startswith("v_"))
sal = sum(c for o, c in cnt.items() if o.startswith("s_"))
print(m.group(1), "loop lines", b - a, "total", tot, "valu", v, "salu", sal, "valu/px", v / 32.0)
for o, c in cnt.most_common(60):
    print(f"{o:28s}{c}")
