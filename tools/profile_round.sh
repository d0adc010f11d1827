# Profiles of one round (GPU box): kernel traces of the bench and of the sampling loop, and three counter passes over tools/traffic_probe.py
# (HBM reads, HBM writes, matrix-core / vector-unit busy cycles).  usage: bash tools/profile_round.sh <tag>
set -e
# (gpurun MERGES what a call wrote into the local gpurun_out/: remove the local copy of prof_<tag> before a re-run, or
#  tools/pmc_summary.py may pick up an older pass's counter file)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sampling -o sampling -- python3 $R/tools/bench_sampling.py --chunk 1250 --steps 20 > $O/sampling_under_rocprof.log 2>&1
export TRAFFIC_MANIFEST=$O/manifest.json
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/traffic_probe.py > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/traffic_probe.py > $O/write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/traffic_probe.py > $O/sq.log 2>&1
ls $O $O/kt
