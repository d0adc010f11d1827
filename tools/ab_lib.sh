# bench.py with several builds of the library in turn (GPU box only):  tools/ab_lib.sh "" mb-vae-dti_amd/variants/libmbvae_x.so ...
# two streams and one stream (where kernel time shows directly), twice each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in "$@"; do
    for mode in "" "--one-stream"; do
      echo -n "${lib:-default} $mode: "
      MBVAE_LIB=$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" || exit 1
    done
  done
done
