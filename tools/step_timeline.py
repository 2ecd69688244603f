"""Kernel timeline of one overlapped train step from a rocprofv3 kernel trace (csv): which queue ran what, when.
usage: step_timeline.py <dir with *_kernel_trace.csv> [step index] [us before] [us after]   (t=0: start of the step's
fold_bn_w1 launch, i.e. the end of its optimizer update)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
before = float(sys.argv[3]) if len(sys.argv) > 3 else 600
after = float(sys.argv[4]) if len(sys.argv) > 4 else 200
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fold = [r for r in rows if 'fold_bn_w1' in r['Kernel_Name']]
t0 = int(fold[k]['Start_Timestamp'])
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 - before * 1000 < s < t0 + after * 1000:
        print(f"{(s - t0) / 1000:8.1f} {(e - t0) / 1000:8.1f} {(e - s) / 1000:7.1f} q{r['Queue_Id']} "
              f"{r['Kernel_Name'].replace('dv::', '').replace('void ', '')[:56]} g{r['Grid_Size_X']}")
