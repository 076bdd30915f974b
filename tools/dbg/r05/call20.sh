# round 5, GPU call 20: closure pass of the build with the self-loading fp8 kernel and the epilogue store loops (tag r05b)
cd $GRAFT_REPO_ROOT
bash tools/closure.sh r05b
