# round 6, GPU call 5: the round-end measurement pass (tools/closure.sh r06)
bash tools/closure.sh r06
