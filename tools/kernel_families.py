import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
fam = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Name"].replace("void ", "").replace("pe::", "").split("(")[0]
    fam[n][0] += int(r["Calls"]); fam[n][1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in fam.values())
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{k[:70]:70s} {v[0]/steps:7.1f} calls/step {v[1]/steps/1e3:9.1f} us/step {v[1]/v[0]/1e3:8.1f} us avg")
print("total", tot/steps/1e3)
