# A/B on one box: residual gradients of the node tensor folded into the products (default) against separate sums
set -e
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
for i in 1 2; do
  echo "default:";              timeout -k 10 200 $B 2>/dev/null | cut -c60-140
  echo "no node fan-out:";      MBVAE_NO_NODE_FANOUT=1 timeout -k 10 200 $B 2>/dev/null | cut -c60-140
  echo "neither (as before):";  MBVAE_NO_NODE_FANOUT=1 MBVAE_NO_GENERIC_ADDEND=1 timeout -k 10 200 $B 2>/dev/null | cut -c60-140
  echo "fan-out, separate sums:"; MBVAE_NO_GENERIC_ADDEND=1 timeout -k 10 200 $B 2>/dev/null | cut -c60-140
done
