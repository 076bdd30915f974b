# round 5, GPU call 7: the whole -m gpu suite on the current build (v3 tiles default, peaked checkpoint, episode replay, graph workspace test)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r05/call07_gputests.txt
