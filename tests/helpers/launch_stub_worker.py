"""Stand-in for a bench.py worker (tests/test_host_cpu.py::test_bench_launcher_*): writes the launcher-provided environment
to <dir>/rank<R>.json, prints one JSON line on stdout from rank 0 only, and leaves with the exit code asked for on its
command line (`--exit R:CODE` for rank R).  Touches neither torch nor the GPU."""
import json
import os
import sys

out_dir = sys.argv[1]
rank = int(os.environ["RANK"])
keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")
with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
    json.dump({"env": {k: os.environ.get(k) for k in keys}, "argv": sys.argv[1:]}, f)
if rank == 0:
    print(json.dumps({"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"])}), flush=True)
code = 0
for i, a in enumerate(sys.argv):
    if a == "--exit":
        r, c = sys.argv[i + 1].split(":")
        if int(r) == rank:
            code = int(c)
sys.exit(code)
