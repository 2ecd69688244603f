"""Print average durations of selected kernels from a rocprofv3 kernel_stats.csv: kstat.py <dir> <substr> [...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in sys.argv[2:]):
        print(r['Name'][:80], r['Calls'], round(float(r['AverageNs']) / 1000, 1), 'us')
