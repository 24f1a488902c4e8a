import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
  n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:50]
  if float(r['Percentage'])>0.5: print('%-52s calls %4s avg %10.1f us  %5s%%'%(n,r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
