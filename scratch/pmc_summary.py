"""Summarise rocprofv3 counter_collection CSVs: mean counter value per kernel."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        if k.startswith("void "): k = k[5:]        # templated kernels carry their return type
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if not k.startswith("auv::"): continue
        print(k, {c: round(sum(v)/len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
