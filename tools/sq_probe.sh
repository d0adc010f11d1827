# One extra counter pass over tools/traffic_probe.py: instruction mix and LDS / memory waits per kernel (GPU box only).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_sq2
rm -rf $O; mkdir -p $O
export TRAFFIC_MANIFEST=$O/manifest.json
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/traffic_probe.py > $O/sq.log 2>&1
ls $O/sq/*/ | head
