#!/bin/bash
# round 4, session t: every workgroup of a pass launch at m = 100 000, on a view and on M (tools/pass_timeline_full.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04u; mkdir -p $O
timeout 600 python tools/pass_timeline_full.py 100000 1 > $O/full_view_m100000.txt 2>&1; echo "view rc=$?" > $O/summary.txt
timeout 600 python tools/pass_timeline_full.py 100000 0 > $O/full_M_m100000.txt 2>&1; echo "M rc=$?" >> $O/summary.txt
timeout 600 python tools/pass_timeline_full.py 300000 1 > $O/full_view_m300000.txt 2>&1; echo "view300k rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/full_view_m100000.txt
