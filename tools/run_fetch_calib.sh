# tools/run_fetch_calib.sh (GPU box, via gpurun): tools/micro/fetch_calib under rocprofv3 --pmc FETCH_SIZE (what the counter reports per access shape)
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- $R/tools/micro/fetch_calib > $OUT/log.txt 2>&1 || { tail $OUT/log.txt; exit 1; }
tail -2 $OUT/log.txt
find $OUT -name "*counter_collection.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
for r in csv.DictReader(open('{}')):
    print(r.get('Kernel_Name','')[:40], r.get('Counter_Name'), r.get('Counter_Value'))
"
