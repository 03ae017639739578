import csv,collections,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ks=sorted(((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],(int(r['Grid_Size_X']),int(r['Grid_Size_Y']),int(r['Grid_Size_Z'])),int(r['Workgroup_Size_X'])) for r in rows))
marks=[i for i,k in enumerate(ks) if 'k_pack_weights_batch' in k[2]]
a,b=marks[10],marks[14]
for pat in sys.argv[2:]:
    c=collections.defaultdict(list)
    for k in ks[a:b]:
        if pat in k[2]: c[(tuple(x//(k[4] if i==0 else 1) for i,x in enumerate(k[3])))].append(k[1]-k[0])
    print(pat, f"total {sum(sum(v) for v in c.values())/4/1e6:.3f} ms/step")
    for g,v in sorted(c.items(), key=lambda kv:-sum(kv[1])):
        print(f"   blocks {g}: {len(v)/4:5.1f}/step avg {sum(v)/len(v)/1e3:7.1f} us total {sum(v)/4/1e6:6.3f} ms")
