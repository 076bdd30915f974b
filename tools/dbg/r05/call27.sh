# round 5, GPU call 27: closure pass of the final build (tag r05c)
cd $GRAFT_REPO_ROOT
bash tools/closure.sh r05c
