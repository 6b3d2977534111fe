import csv, collections, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k[0], k[1], "n=%d" % len(v), "avg=%.4g" % (sum(v) / len(v)), "last=%.4g" % v[-1])
