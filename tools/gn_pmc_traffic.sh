R=$(pwd); O=$R/gpurun_out/gnpmc; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/gn_$c --output-format csv -- python $R/tools/gn_profile_run.py > $O/gn_$c.log 2>&1
done
cd $R; python tools/pmc_summarise.py $(find $O -name "*counter_collection.csv") | grep -v "at::\|rocclr"
