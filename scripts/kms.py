import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print("kernel_ms", round(d["roofline"]["kernel_ms_per_launch"], 4), "frac", round(d["roofline"]["frac"], 4), "fps", round(d["value"]), "ms/step", round(d["ms_per_step"], 4))
