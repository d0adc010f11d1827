# bench.py under one or more option settings (GPU box only):  tools/ab_bench.sh "" "MBVAE_NO_E12_RECOMPUTE=1" ...
cd "$(dirname "$0")/.."
for envs in "$@"; do
  echo "== ${envs:-default}"
  env $envs timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16-line --kernel-table 2> gpurun_out/ab_bench_table.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"
  head -14 gpurun_out/ab_bench_table.txt; grep "sum of HIP" gpurun_out/ab_bench_table.txt
done
