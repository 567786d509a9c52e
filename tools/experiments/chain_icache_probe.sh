export TMPDIR=/tmp REMD_PHASES=1; O=$GRAFT_REPO_ROOT/gpurun_out/r06s3_22; mkdir -p $O; cd /tmp
i=0
for sp in "R R O R R" "V R R O R R V" "O" "R O R"; do
  i=$((i+1)); rm -rf /tmp/pc_$i
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$i -o p -- python $GRAFT_REPO_ROOT/tools/experiments/chain_icache_probe.py "$sp" > $O/run_$i.log 2>&1
  f=$(find /tmp/pc_$i -name "*kernel_stats.csv" | head -1)
  echo "== $sp" >> $O/summary.txt; tail -1 $O/run_$i.log >> $O/summary.txt
  [ -n "$f" ] && python -c "
import csv
for row in csv.reader(open('$f')):
    if row[0]=='Name': continue
    if 'integrate_chain' in row[0] or 'spin_wait' in row[0]: print('%-40s calls %6s avg %8.2f us min %7.2f max %8.2f' % (row[0][:40], row[1], float(row[3])/1e3, float(row[5])/1e3, float(row[6])/1e3))
" >> $O/summary.txt
done
cat $O/summary.txt
