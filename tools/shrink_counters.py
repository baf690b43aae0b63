"""Keep only the columns tools/make_traffic_json.py reads from a rocprofv3 counter-collection CSV (gpurun_out merges at most 64 MiB back)."""
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
keep = ['Kernel_Name', 'Counter_Name', 'Counter_Value', 'Start_Timestamp', 'End_Timestamp']
w = csv.DictWriter(open(p, 'w', newline=''), fieldnames=keep)
w.writeheader()
for r in rows:
    w.writerow({k: r[k] for k in keep})
