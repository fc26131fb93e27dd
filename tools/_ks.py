import csv, sys
tag, pat = sys.argv[1], sys.argv[2]
for B in (1280, 448):
    for r in csv.DictReader(open(f'gpurun_out/r05_e/enc_{B}_kernel_stats.csv')):
        if pat in r['Name']:
            print(tag, B, r['Name'].split('(')[1 if r['Name'].startswith('void (') else 0][:10], r['Name'][28:58], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), round(float(r['MinNs']) / 1e3, 1))
