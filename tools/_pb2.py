import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
pat = sys.argv[1:] or ["wgrad"]
print(d["value"], d["ms_per_step"])
for k in d.get("kernel_breakdown", []):
    if any(p in k.get("kernel", "") for p in pat):
        print(k)
