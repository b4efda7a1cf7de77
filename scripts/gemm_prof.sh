#!/bin/bash
# true kernel durations (rocprofv3 kernel trace) of genrl_sgemm for given "M N K mode" shapes
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for shape in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o g -- python scripts/gemm_one.py $shape > /dev/null 2>&1
  python - "$shape" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open("/tmp/kt/g_kernel_trace.csv")) if "sgemm" in r["Kernel_Name"] or "skinny" in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]]
per = {}
for r in rows:
    k = "reduce" if "splitk" in r["Kernel_Name"] else "gemm"
    per.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
g = per.get("gemm", [0]); rd = per.get("reduce", [0])
grid = [r["Grid_Size_X"] + "x" + r["Grid_Size_Y"] for r in rows if "splitk" not in r["Kernel_Name"]][-1]
print(f"{sys.argv[1]:28s} gemm {min(g):7.1f} us  reduce {min(rd):5.1f} us  grid(threads) {grid}")
PY
done
