# tools/run_spmm_traffic.sh lib.so [lib2.so ...]  (GPU box): FETCH_SIZE and WRITE_SIZE per isolated SpMM launch
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/spmm_traffic; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $OUT/$tag -- python3 $R/tools/spmm_traffic.py "$@" > $OUT/$tag.log 2>&1 || { tail $OUT/$tag.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
res = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "spmm" not in r["Kernel_Name"]: continue
        res.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, (d, v) in enumerate(sorted(res.items())):
    f, w = v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)
    print(f"launch {k:2d}: fetch x2 {2*f*1024/1e6:7.1f} MB  write {w*1024/1e6:6.1f} MB  total {(2*f+w)*1024/1e6:7.1f} MB  L2 hit {v.get('TCC_HIT_sum',0)/max(v.get('TCC_HIT_sum',0)+v.get('TCC_MISS_sum',0),1):.3f}")
PY
