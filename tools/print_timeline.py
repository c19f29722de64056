import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tl/tl_kernel_trace.csv")))
rows = [r for r in rows if "k_" in r["Kernel_Name"][:12]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_scalar" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{nm:18s} q={r['Queue_Id']:>3s} start={s:8.3f} end={e:8.3f} dur={e - s:7.3f} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}")
