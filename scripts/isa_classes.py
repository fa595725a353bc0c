"""Static issue-class split of the chain kernel's main loop for the single-kernel bench configurations, written to
profiles/r03_isa_classes.json (bench.py's roofline.valu.issue_class quotes it through scripts/pmc_merge.py).
Compiles the two instantiation units to ISA (hipcc -S, gfx950) and runs scripts/isa_count.py --json on each kernel.
usage: python scripts/isa_classes.py        (CPU only; ~2 min)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "image_processing_amd", "csrc")
# config -> (instantiation unit, mangled template arguments PRE, POST, RGB, NR, PF, M2, MODE, BAYER)
KERNELS = {
    "c1": ("fused_inst_plain_small.hip", "k_fused_chainILi1ELi1ELb0ELi1ELi3ELb0ELi0ELb0E"),
    "c2": ("fused_inst_plain_small.hip", "k_fused_chainILi1ELi5ELb0ELi1ELi3ELb0ELi0ELb0E"),
    "c4": ("fused_inst_plain_small.hip", "k_fused_chainILi1ELi5ELb0ELi1ELi3ELb0ELi0ELb0E"),
    "c3": ("fused_inst_bayer_small.hip", "k_fused_chainILi1ELi5ELb0ELi1ELi3ELb0ELi0ELb1E"),
}
out, asm = {}, {}
with tempfile.TemporaryDirectory() as tmp:
    for unit in sorted({u for u, _ in KERNELS.values()}):
        asm[unit] = os.path.join(tmp, unit + ".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", asm[unit], os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
    for cfg, (unit, key) in KERNELS.items():
        line = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "isa_count.py"), asm[unit], key, "--json"], text=True)
        out[cfg] = json.loads(line)
out["_what"] = ("per single-kernel bench configuration: vector instructions in the 8-row main loop of its chain kernel (static count, every "
                "branch included) and how many belong to the cheap issue class of profiles/r03_issue_rate_ubench.log; scripts/isa_classes.py")
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_isa_classes.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
