cd "$(dirname "$0")/.."
for rep in 1 2; do for v in 2 4 8 64; do
  echo "== EFB_SYNC=$v"
  MBVAE_EFB_SYNC=$v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16-line --kernel-table 2>&1 >/dev/null | grep -E "fused_bwd|sum of HIP"
done; done
