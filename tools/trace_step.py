"""Print the kernel timeline of one steady-state training step from a rocprofv3 kernel-trace CSV."""
import csv, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'clamp_adam' in r['Kernel_Name']]
a, b = idx[-5], idx[-3]      # two optimizer launches per step (early prefix on the side stream, the rest at the end)
t0 = int(rows[a]['End_Timestamp'])
last_end = t0
for r in rows[a + 1:b + 1]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap={(s - last_end) / 1e3:6.1f} q={r.get('Queue_Id')} grid={r['Grid_Size_X']:>8s} {name}")
    last_end = max(last_end, e)
