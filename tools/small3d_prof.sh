# kernel durations of the 100^3 case (the shipped input_params size) and the two-waves variant beside it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/s3d && mkdir -p $R/gpurun_out/s3d
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s3d -o s3d -- python3 $R/bench.py --nx 100 --ny 100 --nz 100 --steps 2000 --warmup 50 --kernel fused --no-profile --no-cpu-baseline > $R/gpurun_out/s3d/bench.json 2> $R/gpurun_out/s3d/err.log || { tail -5 $R/gpurun_out/s3d/err.log; exit 1; }
cat $R/gpurun_out/s3d/bench.json | cut -c1-200
find $R/gpurun_out/s3d -name "*kernel_stats.csv" | xargs -I{} sh -c 'cut -d, -f1-7 {} | cut -c1-160'
