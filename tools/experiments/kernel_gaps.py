import csv,glob,sys
d=sys.argv[1]
rows=[]
for r in csv.DictReader(open(glob.glob(d+'/*kernel_trace.csv')[0])):
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60]))
rows.sort()
g1=[];g2=[]
for i in range(1,len(rows)):
    s,e,n=rows[i]; ps,pe,pn=rows[i-1]
    if 'resample_mfma' in n and 'pack' in pn: g1.append((s-pe)/1e3)
    if 'dct_quant' in n and 'resample_mfma' in pn: g2.append((s-pe)/1e3)
g1=sorted(g1)[:-2]; g2=sorted(g2)[:-2]
print(d, 'pack->resample median %.1f us, resample->dct median %.1f us' % (g1[len(g1)//2], g2[len(g2)//2]))
