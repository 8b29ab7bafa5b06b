# tools/trace_step.sh (GPU box, via gpurun): rocprofv3 --kernel-trace of bench.py and the kernel timeline of its last step (tools/trace_step.py).
# WL=<bench workload> EXTRA="<more bench flags>" TAG=<suffix> MARKER=<first kernel of a step> PER=<marker launches per step>
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_step
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --workload ${WL:-cora_shape} ${EXTRA:-} > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
DB=$(find $OUT -name "*.db" | head -1)
echo DB=$DB
python3 $R/tools/trace_step.py $DB ${MARKER:-k_row_degree} ${PER:-1} > $R/gpurun_out/trace_${WL:-cora_shape}${TAG:-}_step.txt 2>&1 || tail -5 $R/gpurun_out/trace_${WL:-cora_shape}${TAG:-}_step.txt
rm -rf $OUT
