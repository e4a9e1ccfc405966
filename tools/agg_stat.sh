cd $GRAFT_REPO_ROOT
RFX_LEAF_DBG=128 timeout -k 10 400 python tools/prof_count.py --gbp 18.75 --genome 400000000 --cover 2 --steps 1 2>&1 | grep "record table"
RFX_LEAF_DBG=640 timeout -k 10 400 python tools/prof_count.py --gbp 18.75 --genome 400000000 --cover 2 --steps 1 2>&1 | grep "record table"
