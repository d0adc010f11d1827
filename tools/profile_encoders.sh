# Kernel trace of the encoder-only train step (cfg2), eager, GPU box.  usage: bash tools/profile_encoders.sh <tag>
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_enc_$TAG
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o enc -- python3 $R/tools/bench_configs.py --which encoders --steps 10 --warmup 3 > $O/enc_under_rocprof.log 2>&1
timeout -k 10 200 python3 $R/tools/bench_configs.py --which encoders --steps 20 --warmup 3 > $O/enc_eager.log 2>&1
timeout -k 10 200 python3 $R/tools/bench_configs.py --which encoders --steps 20 --warmup 3 --graph > $O/enc_graph.log 2>&1
tail -2 $O/enc_eager.log $O/enc_graph.log
find $O/kt -name "*kernel_stats.csv" | head
