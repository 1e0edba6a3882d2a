"""Per-launch durations of the dense-layer kernels in the last step of a rocprofv3 kernel trace.
usage: python tools/dense_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'), key=lambda p: -__import__('os').path.getmtime(p))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
keys = sys.argv[2].split(',') if len(sys.argv) > 2 else ('skinny', 'gemm_reduce', 'ProbG')
for r in rows[idx[-2] + 1:idx[-1] + 1]:
    n = r['Kernel_Name']
    if any(k in n for k in keys):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print(f"{d:7.1f} {n[:60]} {(r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])}")
