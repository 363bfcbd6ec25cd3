cd $GRAFT_REPO_ROOT
python tools/full_tiles_ab.py 2>&1 | grep "full tiles"
