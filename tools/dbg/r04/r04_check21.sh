set -x
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
run base
HIP_FORCE_DEV_KERNARG=1 run kernarg1
HIP_FORCE_DEV_KERNARG=0 run kernarg0
ROC_OPT_FLUSH=0 run optflush0
GPU_MAX_HW_QUEUES=8 run hwq8
GPU_MAX_HW_QUEUES=2 run hwq2
HSA_NO_SCRATCH_RECLAIM=1 run noscratchreclaim
ROC_ACTIVE_WAIT_TIMEOUT=1000000 run activewait
DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_LAUNCH_BLOCKING=0 AMD_DIRECT_DISPATCH=1 run directdispatch1
run base2
