side=$1
port=$((20000 + RANDOM % 20000))
for r in 0 1; do
  RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port WORKER_SIDE_STREAM=$side WORKER_OUT=gpurun_out/r3b/ddp_w$side.pt WORKER_HANG_SECONDS=45 timeout -k 5 70 python tests/helpers/ddp_two_streams_worker.py > gpurun_out/r3b/ddp_rank${r}_side$side.log 2>&1 &
done
wait
