import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
for i in range(0,len(d),10):
    print("%4d: "%i + " ".join("%6.1f"%x for x in d[i:i+10]))
