import sys,json
d=json.loads(sys.stdin.read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"]); [print(k) for k in d["kernel_breakdown"] if "chain" in k["kernel"]]
