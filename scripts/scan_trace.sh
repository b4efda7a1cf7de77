cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for b in 64 8; do
python scripts/scan_trace.py $b 2>&1 | grep -v amdgpu
rm -rf /tmp/st; rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o p -- python scripts/scan_trace.py $b > /dev/null 2>&1
python scripts/scan_trace.py --table /tmp/st/p_kernel_trace.csv
done
