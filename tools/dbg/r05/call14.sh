bash tools/closure.sh r05a
