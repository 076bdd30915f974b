set -x
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
run base
COVER_DA_VSPLIT=2 run vsplit2
COVER_DA_VSPLIT=4 run vsplit4
COVER_DA_TAIL=0 run tail0
COVER_SK_SLOTS=256 run skslots256
run base2
