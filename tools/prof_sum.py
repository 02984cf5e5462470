import csv,glob,sys
import os
f=max(glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'), key=os.path.getmtime)
n=int(sys.argv[2]); top=int(sys.argv[3]) if len(sys.argv)>3 else 40
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms/step", round(tot/1e6/n,2))
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs'])/1e6/n:8.2f} ms/step  calls {int(r['Calls'])//n:4d} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
