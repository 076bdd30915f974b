# round 6, GPU call 23: the round-end measurement pass on the final library (MX block scales in)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/closure.sh r06
