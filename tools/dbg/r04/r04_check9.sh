cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_openvla_gpu.py -q -x -k "decode_graph or vision_graph or small_matches" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -k "decision_properties or config2 or config3 or config5" 2>&1 | tail -4
for m in 1 0 1 0; do COVER_DECODE_GRAPH=$m python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DECODE_GRAPH=$m', d['ms_per_step'], d['value'])"; done
