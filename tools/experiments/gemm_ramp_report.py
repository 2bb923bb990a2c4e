import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    name = r["Kernel_Name"]
    if "gemm_nn_kernel<float, 2, 9" not in name and "gemm_tn_kernel<float, 2, 9" not in name:
        prev_end = int(r["End_Timestamp"]); continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e6:10.3f} ms  {'nn' if 'gemm_nn' in name else 'tn'}  dur {(e - s) / 1e3:7.1f} us  gap since previous kernel {((s - prev_end) / 1e3) if prev_end else 0:8.1f} us")
    prev_end = e
