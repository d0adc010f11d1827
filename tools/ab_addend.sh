# A/B on one box: the residual gradient of the node feed-forward block folded into the tiled product's epilogue (default)
# against a separate sum (MBVAE_NO_GENERIC_ADDEND=1).
set -e
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
for i in 1 2; do
  echo "default:";        timeout -k 10 200 $B 2>/dev/null | cut -c60-140
  echo "separate sums:";  MBVAE_NO_GENERIC_ADDEND=1 timeout -k 10 200 $B 2>/dev/null | cut -c60-140
done
