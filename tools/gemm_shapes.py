"""Per-launch-shape GEMM time from a rocprofv3 kernel trace (gpurun_out/prof_ts)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_ts'
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
f = max(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'gemm' in n or 'Cijk' in n:
        short = n.replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')[:64]
        key = (short, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
        agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(k[0].ljust(64), str(k[1:]).ljust(22), '%3d' % len(v), '%8.1f us avg' % (sum(v) / len(v)), '%6.2f ms/step' % (sum(v) / steps / 1e3))
print('all GEMMs: %.2f ms/step' % (sum(sum(v) for v in agg.values()) / steps / 1e3))
