import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k in d.get("kernel_breakdown", d.get("breakdown", []))[:60]:
    if "wgrad" in k.get("kernel", "") or "final" in k.get("kernel", ""):
        print(k)
