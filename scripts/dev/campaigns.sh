# the randomised campaigns on the current tree (round 6: new seeds), one line per campaign in gpurun_out/campaigns/summary.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/campaigns; S=gpurun_out/campaigns/summary.txt; : > $S
run () { name=$1; shift; ( timeout 900 python3 "$@" > gpurun_out/campaigns/$name.log 2>&1; echo "$name rc $? : $(tail -1 gpurun_out/campaigns/$name.log | cut -c1-220)" ) | tee -a $S; }
run engine scripts/dev/fuzz_engine.py 150 606
run pipeline scripts/dev/fuzz_pipeline.py 300 607
run pipeline_randbin scripts/dev/fuzz_pipeline.py 150 608 24000 randbin
run twins_3y scripts/dev/fuzz_twins.py 3y 120 609
run twins_osc scripts/dev/fuzz_twins.py osc 120 610
run fits scripts/dev/fuzz_fits.py 40 611
run misc scripts/dev/fuzz_misc.py 200 612
run ranks scripts/dev/fuzz_ranks.py 12 613
run prob3 scripts/dev/fuzz_prob3.py 300 614
run kde scripts/dev/fuzz_kde.py 200 615
run kde_maps scripts/dev/fuzz_kde_maps.py 60 616
