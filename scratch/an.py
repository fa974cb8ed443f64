import csv,glob,statistics,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
by=collections.OrderedDict()
for r in rows:
    key=(r['Kernel_Name'][:60], r['Grid_Size_X'], r['Workgroup_Size_X'])
    by.setdefault(key,[]).append((int(r['Start_Timestamp']),int(r['End_Timestamp'])))
for k,v in by.items():
    if len(v)<50: continue
    d=[b-a for a,b in v]; per=[v[i+1][0]-v[i][0] for i in range(len(v)-1)]
    print(k, 'n',len(v),'dur med %.2f us'%(statistics.median(d)/1e3),'period med %.2f us'%(statistics.median(per)/1e3))
