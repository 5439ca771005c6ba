# runs the stamped diagnostic build (scripts/micro/i9_stamps.sh, built beforehand) on the GPU box
cd $GRAFT_REPO_ROOT
python scripts/micro/i9_stamps.py 8 512 512 1
python scripts/micro/i9_stamps.py 8 512 512 0
