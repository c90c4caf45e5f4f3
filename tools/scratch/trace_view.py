import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'fit_prologue' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    n=r['Kernel_Name']
    short=n.split('(')[0][-50:]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} q{r['Queue_Id']} g{r['Grid_Size_X']}x{r['Grid_Size_Z']} {short}")
