import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
from collections import defaultdict
d=defaultdict(list)
for r in csv.DictReader(open(f)):
    if "wgrad_tile16" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:45]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print(k, ["%.0f"%x for x in v])
