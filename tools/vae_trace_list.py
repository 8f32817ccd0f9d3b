import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "conv_kernel" in r["Kernel_Name"] or "ncl" in r["Kernel_Name"] or "peak" in r["Kernel_Name"] or "absmax" in r["Kernel_Name"]]
rows = rows[len(rows)//2:]
tot = 0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = r["Kernel_Name"].replace("void ace355::(anonymous namespace)::", "").split("(")[0]
    print(f"{d:9.1f} us grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) wg={r['Workgroup_Size_X']} lds={r.get('LDS_Block_Size','?')} {name}")
print("total ms", tot / 1e3)
