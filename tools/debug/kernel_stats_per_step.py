import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
steps=float(sys.argv[2])
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows[:24]:
    per=float(r['TotalDurationNs'])/steps/1000
    print(f"{per:8.1f} us/step  calls/step {int(r['Calls'])/steps:.1f}  {r['Name'][:80]}")
print("sum", sum(float(r['TotalDurationNs']) for r in rows)/steps/1000)
