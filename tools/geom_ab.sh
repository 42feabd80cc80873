
mkdir -p gpurun_out/geomab
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in cur wg128 wg256 empty empty256; do
  if [ $v = product ]; then export BEVWARP_LIB=$R/bev_amd/csrc/libbevwarp.so; else export BEVWARP_LIB=$R/bev_amd/csrc/variants/$v.so; fi
  python3 $R/tools/time_tracker.py > $R/gpurun_out/geomab/$v.time 2>&1
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/tools/time_tracker.py > /tmp/prof_$v.log 2>&1 || tail -5 /tmp/prof_$v.log
  echo "== $v"; cat $R/gpurun_out/geomab/$v.time
  python3 - $v <<'PY'
import csv,glob,sys
f=glob.glob('/tmp/prof_%s/**/*kernel_stats.csv'%sys.argv[1],recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'iou' in r['Name'] or 'tracker' in r['Name']:
        print('   %-60s calls %s avg %.2f us min %.2f'%(r['Name'][:60],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3))
PY
done
