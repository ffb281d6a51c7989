# wall time of the driver on the shipped input_params (100^3, 3 steps, all nine fields written every step)
R=$GRAFT_REPO_ROOT
rm -rf /tmp/run1 && mkdir -p /tmp/run1/output && cp -r $R/data /tmp/run1/ && cd /tmp/run1 || exit 1
s=$(date +%s.%N)
timeout -k 10 120 $R/fdtd_amd/fdtd --no-materials > log.txt 2> err.txt; rc=$?
e=$(date +%s.%N)
tail -2 log.txt; tail -2 err.txt
echo "rc=$rc files=$(ls output | wc -l) bytes=$(du -sb output | cut -f1) seconds=$(echo "$e - $s" | bc)"
