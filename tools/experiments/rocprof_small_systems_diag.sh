export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06s3_20; mkdir -p $O; cd /tmp
for c in hv; do
  rm -rf /tmp/prof_$c
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -o p -- python -X faulthandler -c "
import faulthandler, sys; faulthandler.dump_traceback_later(80, exit=True)
sys.argv = ['x', '$c']; sys.path.insert(0, '$GRAFT_REPO_ROOT'); sys.path.insert(0, '$GRAFT_REPO_ROOT/tools')
import bench_configs; bench_configs.main()
" > $O/run_$c.log 2>&1
  echo "rc $c $?" >> $O/status.txt
  f=$(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python -c "
import csv,sys
for row in csv.reader(open('$f')):
    if row[0]=='Name': continue
    print('%-44s calls %6s total %9.2f ms avg %8.2f us min %7.2f max %8.2f' % (row[0][:44], row[1], float(row[2])/1e6, float(row[3])/1e3, float(row[5])/1e3, float(row[6])/1e3))
" | head -14 > $O/kernel_stats_$c.csv
done
cat $O/status.txt; cat $O/kernel_stats_hv.csv
