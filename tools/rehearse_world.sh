# tools/rehearse_world.sh (GPU box, via gpurun): the sharded GPU tests, then one rank's share of a G-rank step for G = 2, 4, 8 and both tower
# splits (bench.py --rehearse-world, collectives replaced by no-ops), the 1-rank RCCL path and the other --workload lines -> gpurun_out/r3_*
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_sharded.py "tests/test_gpu_parity.py::test_eight_rank_shards_reproduce_single_gpu_step_at_amazon_computer_size" -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_tests_c.log 2>&1 || { tail -30 gpurun_out/r3_tests_c.log; exit 1; }
tail -3 gpurun_out/r3_tests_c.log
: > gpurun_out/r3_rehearse.jsonl
for G in 2 4 8; do
  for REP in "" "--replicate-graph-part"; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 --rehearse-world $G $REP >> gpurun_out/r3_rehearse.jsonl 2>> gpurun_out/r3_rehearse.err
  done
done
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --rehearse-world 8 --rehearse-rank 7 >> gpurun_out/r3_rehearse.jsonl 2>> gpurun_out/r3_rehearse.err
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --rehearse-world 8 --rehearse-rank 7 --replicate-graph-part >> gpurun_out/r3_rehearse.jsonl 2>> gpurun_out/r3_rehearse.err
timeout -k 10 200 python bench.py --force-plan --no-cpu-baseline --no-extras > gpurun_out/r3_forceplan.json 2>> gpurun_out/r3_rehearse.err
timeout -k 10 200 python bench.py --force-plan --replicate-graph-part --no-cpu-baseline --no-extras > gpurun_out/r3_forceplan_rep.json 2>> gpurun_out/r3_rehearse.err
timeout -k 10 200 python bench.py --workload amazon_photo --no-cpu-baseline > gpurun_out/r3_bench_photo.json 2>> gpurun_out/r3_rehearse.err
timeout -k 10 200 python bench.py --workload cora_async > gpurun_out/r3_bench_cora_async.json 2>> gpurun_out/r3_rehearse.err
echo all done
