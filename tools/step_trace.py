"""Print the kernel sequence of ONE steady-state optimizer step from a rocprofv3 kernel trace
(<dir>/*_kernel_trace.csv): name, grid (workgroups), duration, gap to the previous kernel."""
import csv, glob, sys, re
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"pvae::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")[:70]
# a step starts at the kernel that follows the last launch of the previous step (wgrad_pair_kernel)
idx = [i for i, r in enumerate(rows) if "wgrad_pair_kernel" in r["Kernel_Name"] or "wgrad_pair_gather_kernel" in r["Kernel_Name"]]
# of the steady-state steps (second half of the trace) the one with the shortest wall time: under the tracer the host
# sometimes falls behind the device for a few launches, which shows as gaps that an un-traced run does not have
def wall(k):
    return int(rows[idx[k + 1]]["End_Timestamp"]) - int(rows[idx[k] + 1]["Start_Timestamp"])
best = min(range(len(idx) // 2, len(idx) - 1), key=wall)
lo, hi = idx[best] + 1, idx[best + 1] + 1
prev_end, t0, tot = None, int(rows[lo]["Start_Timestamp"]), 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    print("%7.2f us  +%5.2f gap  %5d wg  %s" % ((e - s) / 1e3, 0 if prev_end is None else (s - prev_end) / 1e3, wg, short(r["Kernel_Name"])))
    prev_end = e; tot += e - s
print("step: %.2f us wall, %.2f us in kernels, %d launches" % ((prev_end - t0) / 1e3, tot / 1e3, hi - lo))
