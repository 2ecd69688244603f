"""Per-queue view of one overlapped train step from a rocprofv3 kernel-trace csv: busy time per queue, time with no kernel on
any queue, time with exactly one / two / three queues busy, and the main queue's kernels with the gap in front of each and
what the other queues ran during that gap.
usage: queue_view.py <dir> [step index] [marker kernel substring: fold_bn_w1 (fp32) | bf_cast (bf16)] [min gap us]"""
import csv, glob, sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
marker = sys.argv[3] if len(sys.argv) > 3 else 'fold_bn_w1'
ming = float(sys.argv[4]) if len(sys.argv) > 4 else 8
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [r for r in rows if marker in r['Kernel_Name']]
t0, t1 = int(marks[k]['Start_Timestamp']), int(marks[k + 1]['Start_Timestamp'])
step = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
print(f"step {k}: {(t1 - t0) / 1000:.1f} us, {len(step)} kernels")
busy = {}
for r in step:
    busy[r['Queue_Id']] = busy.get(r['Queue_Id'], 0) + int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("busy us per queue:", {q: round(v / 1000, 1) for q, v in sorted(busy.items())})
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']), 1))
    ev.append((min(int(r['End_Timestamp']), t1), -1))
ev.sort()
depth, last, hist = 0, t0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + t - last
    last = t
    depth += d
print("us with n kernels in flight:", {n: round(v / 1000, 1) for n, v in sorted(hist.items())})
mq = marks[k]['Queue_Id']
main = [r for r in step if r['Queue_Id'] == mq]
name = lambda r: r['Kernel_Name'].replace('dv::', '').replace('void ', '')[:34]
prev = None
for r in main:
    s = int(r['Start_Timestamp'])
    if prev is not None:
        pe = int(prev['End_Timestamp'])
        g = (s - pe) / 1000
        if g >= ming:
            others = [o for o in step if o['Queue_Id'] != mq and int(o['Start_Timestamp']) < s and int(o['End_Timestamp']) > pe]
            print(f"  gap {g:6.1f} us at {(s - t0) / 1000:8.1f}: {name(prev)} -> {name(r)} | meanwhile: " +
                  ", ".join(f"q{o['Queue_Id']} {name(o)} ({(int(o['End_Timestamp']) - int(o['Start_Timestamp'])) / 1000:.0f})" for o in others[:4]))
    prev = r
