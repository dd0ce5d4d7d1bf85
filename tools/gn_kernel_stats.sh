#!/bin/bash
# average duration of every GN kernel of the C2 solve alone (tools/gn_profile_run.py under rocprofv3 --kernel-trace --stats); CGMR_LIB selects the build
R=$(pwd); O=/tmp/gnstats_$$; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tools/gn_profile_run.py > $O/log 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if n.startswith('k_') or 'cgmr' in n: print('%-34s calls %5s avg %8.2f us' % (n[:34], r['Calls'], float(r['AverageNs'])/1e3))
"
