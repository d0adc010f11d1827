set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01e
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o bench -- python3 $R/bench.py --steps 5 --warmup 2 > $O/bench_under_rocprof.log 2>&1
export TRAFFIC_MANIFEST=$O/manifest.json
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/traffic_probe.py > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/traffic_probe.py > $O/write.log 2>&1
ls $O $O/kt
