# bench.py kernel table lines matching a pattern under several settings of one option (same box):
#   tools/ab_opt.sh OPTION "v1 v2 ..." 'grep pattern'
cd "$(dirname "$0")/.."
for rep in 1 2; do for v in $2; do
  echo "== $1=$v"
  env MBVAE_$1=$v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16-line --kernel-table 2>&1 >/dev/null | grep -E "$3|sum of HIP"
done; done
