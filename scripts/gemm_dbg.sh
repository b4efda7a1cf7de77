#!/bin/bash
cd $GRAFT_REPO_ROOT
SRC="genrl_amd/csrc/gemm.hip genrl_amd/csrc/rowops.hip genrl_amd/csrc/dist.hip genrl_amd/csrc/conv.hip genrl_amd/csrc/optim.hip"
for flags in "" "-DGENRL_DBG_NO_STORE" "-DGENRL_DBG_NO_EPILOGUE"; do
  out=/tmp/lib_dbg.so; rm -f $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $flags -o $out $SRC 2>&1 | grep -E "error" -A3
  echo "=== flags: $flags"
  for K in 64; do
    cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
    GENRL_HIP_SO=$out rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o g -- python scripts/gemm_one.py 1024 1024 $K kk > /dev/null 2>&1
    python - <<PY
import csv
rows = [r for r in csv.DictReader(open("/tmp/kt/g_kernel_trace.csv")) if "sgemm" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
print("K=$K durations us", [round(x,1) for x in d])
PY
  done
done
