# Counter pass over tools/bwd_both_probe.py (GPU box only): where the single-pass projection backward spends its wave cycles
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_pb
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/bwd_both_probe.py > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/sq/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if "proj_bwd" in k or "tswgrad_reduce" in k:
        print(k, {a: f"{b:.3g}" for a, b in d.items()})
PY
