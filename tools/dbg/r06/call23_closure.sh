# round 6, GPU calls 23 and 29: the round-end measurement pass on the final library (call 23: MX block scales in; call 29: + the shared-keys decode attention)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/closure.sh r06
