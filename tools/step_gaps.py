"""Idle gaps of the main queue and per-queue busy time over one overlapped train step (rocprofv3 kernel trace csv).
usage: step_gaps.py <dir> [step index] [min gap us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ming = float(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fold = [r for r in rows if 'fold_bn_w1' in r['Kernel_Name']]
t0, t1 = int(fold[k]['Start_Timestamp']), int(fold[k + 1]['Start_Timestamp'])
step = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
print(f"step {k}: {(t1 - t0) / 1000:.1f} us, {len(step)} kernels")
busy = {}
for r in step:
    busy[r['Queue_Id']] = busy.get(r['Queue_Id'], 0) + int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("busy us per queue:", {q: round(v / 1000, 1) for q, v in sorted(busy.items())})
# union of all queues: time with no kernel at all
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
cur_end, idle = t0, 0
for s, e in ev:
    if s > cur_end:
        idle += s - cur_end
    cur_end = max(cur_end, e)
print(f"no kernel running on any queue: {idle / 1000:.1f} us")
mq = fold[k]['Queue_Id']
main = [r for r in step if r['Queue_Id'] == mq]
prev = None
tot = 0
for r in main:
    s = int(r['Start_Timestamp'])
    if prev is not None:
        g = (s - int(prev['End_Timestamp'])) / 1000
        if g >= ming:
            tot += g
            print(f"  gap {g:6.1f} us at {(s - t0) / 1000:8.1f}: {prev['Kernel_Name'].replace('dv::','')[:40]} -> {r['Kernel_Name'].replace('dv::','')[:40]}")
    prev = r
print(f"main-queue gaps >= {ming} us: {tot:.1f} us")
