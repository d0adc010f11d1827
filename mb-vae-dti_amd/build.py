"""Build libmbvae_hip.so (gfx950) in-tree with hipcc.  `python mb-vae-dti_amd/build.py [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmbvae_hip.so")
SOURCES = ["gemm.hip", "edge_attn.hip", "edge_fused.hip", "rowops.hip", "loss.hip", "diffusion.hip", "eigen.hip", "cycles.hip"]
# No packed-fp32 instructions (v_pk_*_f32) in device code.  Measured on MI355X: the node-path LayerNorm backward
# (ln_res_bwd_kernel<64, 1>), whose compiled form reads its per-row statistics through op_sel'ed packed operands, returned
# rare wrong rows (the low half of one packed subtract behaving as "minus 0" in lanes 48-63) whenever tall-skinny product
# kernels of ANOTHER stream were running on the chip, and never on an idle one; compiled without packed fp32 the same
# source is bit-stable next to them (tools/concurrency_probe.py, tests/test_gpu_concurrency.py).  The train step's time is
# the same with and without (54.5-54.9 ms either way): the row kernels are memory bound and the matrix kernels' epilogues
# do not depend on packed issue.  The host pass does not know the feature and says so; that warning is filtered below.
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=on",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [os.path.join(CSRC, "common.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + common):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        err = "\n".join(l for l in res.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l)
        if err.strip():
            print(err, file=sys.stderr, flush=True)
        if res.returncode != 0:
            raise subprocess.CalledProcessError(res.returncode, cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT, *objs])
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
