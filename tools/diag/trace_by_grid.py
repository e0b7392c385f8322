import csv,sys,collections
tr=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in tr:
    k=(r['Kernel_Name'][:34], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'), r.get('LDS_Block_Size'))
    d[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()):
    if len(v)>3: print(k, len(v), round(sum(v)/len(v)/1e3,1),'us  min',round(min(v)/1e3,1))
