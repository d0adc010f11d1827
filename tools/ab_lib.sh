# bench.py with several builds of the library in turn (GPU box only):  tools/ab_lib.sh "" mb-vae-dti_amd/variants/libmbvae_x.so ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for lib in "$@"; do
  echo "== ${lib:-default}"
  for rep in 1 2; do
    MBVAE_LIB=$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-line | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" || exit 1
  done
done
